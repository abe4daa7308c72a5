# round 5 (third session): cc2d on the column-block-major layout -- 512-thread column workgroups, pass sizes
set -o pipefail
rm -f gpurun_out/q11_ab.log
for rep in 1 2 3; do
  for l in libxcx.so libxcx_n5w4.so libxcx_n5w6.so; do
    XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q11_ab.log || exit 1
  done
  for m in 512 2048 4096; do
    XCX_CC2D_PASS_MB=$m timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('libxcx.so PASS_MB=$m', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q11_ab.log
  done
done
XCX_CC2D_OVERLAP=0 timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('libxcx.so OVERLAP=0', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q11_ab.log
cat gpurun_out/q11_ab.log
