# round 5 (third session): cc2d -- S2 column-block-major, T2 row-group sweep
set -o pipefail
rm -f gpurun_out/q9_ab.log
for rep in 1 2 3; do
  for r in 1 2 4 8 16 32 512; do
    XCX_LIB_PATH=$PWD/xcorr_amd/lib/libxcx_t$r.so timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T2_RB=$r', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q9_ab.log || exit 1
  done
done
cat gpurun_out/q9_ab.log
for r in 1 4 16; do
XCX_LIB_PATH=$PWD/xcorr_amd/lib/libxcx_t$r.so XCX_CC2D_OVERLAP=0 tools/pmc_cc2d.sh q9_t$r > gpurun_out/q9_pmc_t$r.log 2>&1; tail -8 gpurun_out/q9_pmc_t$r.log | cut -c1-40,180-260
done
