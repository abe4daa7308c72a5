#!/bin/bash
# counters of every kernel of an arbitrary command: tools/pmc_any.sh <tag> "<counters>" <python script> [args]
set -o pipefail
tag=$1; ctrs=$2; shift; shift
root=$(pwd); out=$root/gpurun_out/pmc_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctrs --output-format csv -d $out/run -- python3 $root/"$@" > $out/run.log 2>&1 || { tail -20 $out/run.log; exit 1; }
cd $root
python3 - $out <<'PY'
import csv, glob, sys, os, re
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob(os.path.join(out, 'run', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = cnt[(k, c)]
        print(f'   {c:28s} avg/launch {acc[k][c]/n:16.1f}   launches {n}')
PY
