# round 5 (second session): forward transforms stream (nt) when their Tf is beyond the cache -- suite + A/B
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5o_pytest.txt 2>&1; tail -3 gpurun_out/r5o_pytest.txt
tools/envsweep.sh 3 "XCX_FWD_NT=0" "XCX_FWD_NT=1" "XCX_X=1" > gpurun_out/r5o_sweep.txt 2>&1
cat gpurun_out/r5o_sweep.txt
for v in 0 1; do XCX_FWD_NT=$v timeout -k 10 300 python bench.py --stack --days 30 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stack nt=$v', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernels_us'])"; done > gpurun_out/r5o_stack.txt 2>&1
cat gpurun_out/r5o_stack.txt
