#!/bin/bash
# kernel trace + counter passes of the 2-D pair pass (tools/bench_timelapse.py, one pass at a time: XCX_CC2D_OVERLAP=0)
# usage: tools/pmc_cc2d.sh <tag>
set -o pipefail
tag=$1
root=$(pwd); out=$root/gpurun_out/cc2d_$tag; mkdir -p $out
export TMPDIR=/tmp XCX_CC2D_OVERLAP=${XCX_CC2D_OVERLAP:-0}; cd /tmp
B="$root/tools/bench_timelapse.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $B > $out/kt.log 2>&1 || { tail -20 $out/kt.log; exit 1; }
find $out/kt -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $B > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; echo "pass $i ($ctrs) failed"; continue; }
done
cd $root
python3 - $out <<'PY'
import csv, glob, sys, os, re
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k2d_[a-z_]+(<[^>]*>)?)', r['Kernel_Name'])
        if not m: continue
        k = m.group(1)[:40] + (' inv' if 'true>(' in r['Kernel_Name'] or 'Lb1' in r['Kernel_Name'] else '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
with open(os.path.join(out, 'counters.txt'), 'w') as fo:
    for k in sorted(acc):
        print(k, file=fo)
        for c in sorted(acc[k]):
            n = cnt[(k, c)]
            print(f'   {c:28s} avg/launch {acc[k][c]/n:18.1f}   launches {n}', file=fo)
print(open(os.path.join(out, 'counters.txt')).read())
PY
cut -c1-60,200- $out/kernel_stats.csv | head -8
rm -rf $out/kt $out/p?
