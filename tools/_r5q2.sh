# round 5 (third session): cc2d register budgets (column pass at 4 waves per SIMD, row pass at 6) and the row pass's
# first stage with prefetched untangle twiddles -- interleaved A/B of library builds on one box
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_cc2d.py -x -q > gpurun_out/q2_pytest.log 2>&1; rc=$?; tail -2 gpurun_out/q2_pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/check_cc2d.py 30 11 > gpurun_out/q2_check_main.log 2>&1 || { tail -5 gpurun_out/q2_check_main.log; exit 1; }
XCX_CC2D_PASS_MB=1 XCX_LIB_PATH=$PWD/xcorr_amd/lib/libxcx_c4.so timeout -k 10 200 python tools/check_cc2d.py 30 12 > gpurun_out/q2_check_c4.log 2>&1 || { tail -5 gpurun_out/q2_check_c4.log; exit 1; }
tail -q -n 1 gpurun_out/q2_check_*.log
for rep in 1 2 3; do
  for l in libxcx_tw0.so libxcx.so libxcx_c4.so libxcx_c4tw0.so libxcx_r6.so libxcx_c4r6.so; do
    XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q2_ab.log || exit 1
  done
done
XCX_CC2D_OVERLAP=0 XCX_LIB_PATH=$PWD/xcorr_amd/lib/libxcx_c4.so timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 OVERLAP=0', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q2_ab.log
XCX_CC2D_OVERLAP=0 timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('main OVERLAP=0', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q2_ab.log
cat gpurun_out/q2_ab.log
