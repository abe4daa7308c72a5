# round 5 (third session): cc2d -- spectra S2 and intermediate T2 both in row groups; RB sweep
set -o pipefail
for l in libxcx_rb32.so libxcx_rb512.so; do
XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 300 python -m pytest tests/test_gpu_cc2d.py tests/test_gpu_spectrogram.py -x -q > gpurun_out/q7_pytest_$l.log 2>&1; rc=$?; tail -2 gpurun_out/q7_pytest_$l.log
[ $rc -ne 0 ] && exit 1
XCX_CC2D_PASS_MB=64 XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python tools/check_cc2d.py 25 51 > gpurun_out/q7_check_$l.log 2>&1 || { tail -5 gpurun_out/q7_check_$l.log; exit 1; }
done
tail -q -n 1 gpurun_out/q7_check_*.log
rm -f gpurun_out/q7_ab.log
for rep in 1 2 3; do
  for l in libxcx_rb1.so libxcx_rb16.so libxcx_rb32.so libxcx_rb64.so libxcx_rb512.so; do
    XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q7_ab.log || exit 1
  done
done
cat gpurun_out/q7_ab.log
