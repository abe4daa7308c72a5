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
python3 - $out $root <<'PY'
import csv, glob, hashlib, json, sys, os, re
from collections import defaultdict
out, root = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k2d_[a-z_]+)(<.*>)?', r['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        if k == 'k2d_cols':                                  # template arguments: <Radices<...>, INV, threads>
            k += '<inv>' if re.search(r'>, true', r['Kernel_Name']) else '<fwd>'
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
with open(os.path.join(out, 'counters.txt'), 'w') as fo:
    for k in sorted(acc):
        print(k, file=fo)
        for c in sorted(acc[k]):
            n = cnt[(k, c)]
            print(f'   {c:28s} avg/launch {acc[k][c]/n:18.1f}   launches {n}', file=fo)
print(open(os.path.join(out, 'counters.txt')).read())
# HBM bytes per tile pair, stamped with the library's hash (tools/bench_timelapse.py fills roofline.traffic from it when
# the hash matches): FETCH_SIZE / WRITE_SIZE are KiB, fetch x 2 on gfx950; 3 runs of 2016 pairs (--steps 2 --warmup 1)
pairs = 3 * 2016.0
def per_pair(k, c, f=1.0): return acc[k].get(c, 0.0) * 1024.0 * f / pairs
if 'FETCH_SIZE' in acc.get('k2d_cols<inv>', {}) and 'WRITE_SIZE' in acc.get('k2d_cols<inv>', {}):
    sys.path.insert(0, root)
    import bench
    lib = os.environ.get('XCX_LIB_PATH') or os.path.join(root, 'xcorr_amd', 'lib', 'libxcx.so')
    d = {'_stamp': {'tag': os.path.basename(out).replace('cc2d_', ''), 'lib_sha256': hashlib.sha256(open(lib, 'rb').read()).hexdigest(),
                    'src_sha256': bench.src_sha256(), 'workload': 'configs[4] timelapse: 64 tiles 256x4096 -> 2016 tile pairs',
                    'source': 'tools/pmc_cc2d.sh: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE (KiB; fetch x 2 on gfx950)'},
         'k2d_cols_inv': {'fetch_bytes_per_tile_pair_corrected_x2': per_pair('k2d_cols<inv>', 'FETCH_SIZE', 2.0),
                          'write_bytes_per_tile_pair': per_pair('k2d_cols<inv>', 'WRITE_SIZE')},
         'k2d_rows_inv': {'fetch_bytes_per_tile_pair_corrected_x2': per_pair('k2d_rows_inv', 'FETCH_SIZE', 2.0),
                          'write_bytes_per_tile_pair': per_pair('k2d_rows_inv', 'WRITE_SIZE')},
         'tile_transforms_bytes_per_tile_pair': sum(per_pair(k, 'FETCH_SIZE', 2.0) + per_pair(k, 'WRITE_SIZE') for k in ('k2d_rows_fwd', 'k2d_cols<fwd>'))}
    d['hbm_bytes_per_tile_pair'] = (sum(v for k in ('k2d_cols_inv', 'k2d_rows_inv') for v in d[k].values()) + d['tile_transforms_bytes_per_tile_pair'])
    json.dump(d, open(os.path.join(out, 'cc2d_pmc_latest.json'), 'w'), indent=1)
    print('cc2d_pmc_latest.json:', round(d['hbm_bytes_per_tile_pair'] / 1e6, 2), 'MB per tile pair  (copy it to profiles/ with the run\'s other files)')
PY
cut -c1-60,200- $out/kernel_stats.csv | head -8
rm -rf $out/kt $out/p?
