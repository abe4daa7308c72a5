#!/bin/bash
# SQ counters of the pair kernels on a reduced bench. usage: tools/pmc.sh <tag> "<counters>" [bench args]
set -o pipefail
tag=$1; ctrs=$2; shift; shift
root=$(pwd); out=$root/gpurun_out/pmc_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc $ctrs --output-format csv -d $out/run -- python3 $root/bench.py --stations 16 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing "$@" > $out/bench.log 2>&1 || { tail -20 $out/bench.log; exit 1; }
cd $root
python3 - $out <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob(os.path.join(out, 'run', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for s in ('k_wpair_dual', 'k_gram', 'k_wpair_rows', 'k_wpair_cols', 'k_wfwd_rows', 'k_wfwd_cols', 'k_pair_rows', 'k_pair_cols'):
            if s in k: k = s
        if not k.startswith('k_'): continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = cnt[(k, c)]
        print(f'   {c:28s} avg/launch {acc[k][c]/n:16.1f}   launches {n}')
PY
