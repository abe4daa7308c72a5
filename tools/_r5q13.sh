# round 5 (third session): cc2d tile groups of 8 x 8 / 4 x 8 tiles (0.25 / 0.375 instead of 0.5 slab reads per pair across the fabric)
set -o pipefail
for l in libxcx_g88.so; do
XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 300 python -m pytest tests/test_gpu_cc2d.py -x -q > gpurun_out/q13_pytest.log 2>&1; rc=$?; tail -2 gpurun_out/q13_pytest.log
[ $rc -ne 0 ] && exit 1
done
rm -f gpurun_out/q13_ab.log
for rep in 1 2 3; do
  for cfg in "libxcx.so 1024" "libxcx_g88.so 1024" "libxcx_g88.so 2200" "libxcx_g48.so 1100" "libxcx_g84.so 1100" "libxcx_g48.so 2200"; do
    set -- $cfg
    XCX_CC2D_PASS_MB=$2 XCX_LIB_PATH=$PWD/xcorr_amd/lib/$1 timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 PASS_MB=$2', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q13_ab.log || exit 1
  done
done
cat gpurun_out/q13_ab.log
XCX_LIB_PATH=$PWD/xcorr_amd/lib/libxcx_g88.so XCX_CC2D_PASS_MB=2200 XCX_CC2D_OVERLAP=0 tools/pmc_cc2d.sh q13_g88 > gpurun_out/q13_pmc.log 2>&1; grep -A3 "k2d_cols<xcx::Radices<8, 8, 8, 1> inv" gpurun_out/cc2d_q13_g88/counters.txt | head -3; grep "WRITE_SIZE" gpurun_out/cc2d_q13_g88/counters.txt | head -3; awk -F'","|",|,' '/k2d_rows_inv|k2d_cols/{print substr($1,1,50), $2, $(NF-4)}' gpurun_out/cc2d_q13_g88/kernel_stats.csv | head -2
