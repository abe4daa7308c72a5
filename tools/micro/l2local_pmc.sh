#!/bin/bash
# fabric-side counters of tools/micro/l2local (run on the GPU box): one pass per counter
set -o pipefail
root=$(pwd); out=$root/gpurun_out/l2local; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d $out/$c -- $root/tools/micro/l2local ${1:-2} ${2:-2000} > $out/$c.log 2>&1 || { tail $out/$c.log; exit 1; }
done
cd $root
python3 - $out <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
for c in ('WRITE_SIZE', 'FETCH_SIZE'):
    rows = []
    for f in glob.glob(os.path.join(out, c, '**', '*counter_collection.csv'), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if 'l2local' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    print(c, [round(float(r['Counter_Value'])) for r in rows])
PY
