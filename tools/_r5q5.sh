# round 5 (third session): cc2d column workgroups of 512 threads (two butterflies per thread: ~90 registers, one load batch)
set -o pipefail
for l in libxcx_n5w4.so libxcx_n5w6.so; do
XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 300 python -m pytest tests/test_gpu_cc2d.py -x -q > gpurun_out/q5_pytest_$l.log 2>&1; rc=$?; tail -2 gpurun_out/q5_pytest_$l.log
[ $rc -ne 0 ] && exit 1
XCX_CC2D_PASS_MB=64 XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python tools/check_cc2d.py 25 31 > gpurun_out/q5_check_$l.log 2>&1 || { tail -5 gpurun_out/q5_check_$l.log; exit 1; }
done
tail -q -n 1 gpurun_out/q5_check_*.log
rm -f gpurun_out/q5_ab.log
for rep in 1 2 3; do
  for l in libxcx.so libxcx_n5w4.so libxcx_n5w6.so; do
    XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q5_ab.log || exit 1
  done
done
for l in libxcx_n5w4.so libxcx_n5w6.so; do
XCX_CC2D_OVERLAP=0 XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l OVERLAP=0', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q5_ab.log
done
cat gpurun_out/q5_ab.log
