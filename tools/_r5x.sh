# round 5 (second session): forward row kernel without its padding loads -- parity + A/B
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config3.py tests/test_gpu_config2.py tests/test_gpu_config4.py -x -q -m gpu > gpurun_out/r5x_pytest.txt 2>&1; tail -3 gpurun_out/r5x_pytest.txt
tools/envsweep.sh 3 "XCX_FWD_HALFPAD=0" "XCX_FWD_HALFPAD=1" > gpurun_out/r5x_sweep.txt 2>&1
cat gpurun_out/r5x_sweep.txt
for v in 0 1; do XCX_FWD_HALFPAD=$v timeout -k 10 300 python bench.py --stack --days 30 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stack halfpad=$v', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernels_us'])"; done > gpurun_out/r5x_stack.txt 2>&1
cat gpurun_out/r5x_stack.txt
