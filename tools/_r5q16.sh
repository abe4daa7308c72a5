# round 5 (third session): Gram kernel -- old accumulator values fetched up front in the result phase (A/B against the r05k build)
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_config4.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -k "stack or gram or config4 or reduce" > gpurun_out/q16_pytest.log 2>&1; rc=$?; tail -2 gpurun_out/q16_pytest.log
[ $rc -ne 0 ] && exit 1
rm -f gpurun_out/q16_ab.log
for rep in 1 2 3; do
  for l in libxcx_gramold.so libxcx.so; do
    XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 300 python bench.py --stack --days 30 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['ms_per_step'],2), round(d['roofline']['kernels_us']['k_gram'],1))" >> gpurun_out/q16_ab.log || exit 1
  done
done
cat gpurun_out/q16_ab.log
