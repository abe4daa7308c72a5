# round 5 (third session): cc2d -- row kernel with the value-only arg-max, two passes in flight; A/B of the overlap
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_cc2d.py -x -q > gpurun_out/q1_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/q1_pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/check_cc2d.py 40 1 > gpurun_out/q1_check_a.log 2>&1 || { tail -5 gpurun_out/q1_check_a.log; exit 1; }
XCX_CC2D_PASS_MB=1 timeout -k 10 200 python tools/check_cc2d.py 40 2 > gpurun_out/q1_check_b.log 2>&1 || { tail -5 gpurun_out/q1_check_b.log; exit 1; }
XCX_CC2D_OVERLAP=0 XCX_CC2D_PASS_MB=1 timeout -k 10 200 python tools/check_cc2d.py 25 3 > gpurun_out/q1_check_c.log 2>&1 || { tail -5 gpurun_out/q1_check_c.log; exit 1; }
tail -q -n 1 gpurun_out/q1_check_?.log
for rep in 1 2; do
  for mode in "0 1024" "1 1024" "1 512" "1 256" "1 2048"; do
    set -- $mode
    echo "OVERLAP=$1 PASS_MB=$2" >> gpurun_out/q1_ab.log
    XCX_CC2D_OVERLAP=$1 XCX_CC2D_PASS_MB=$2 timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q1_ab.log || exit 1
  done
done
cat gpurun_out/q1_ab.log
