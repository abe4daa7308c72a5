timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5f_pytest.txt 2>&1
tail -6 gpurun_out/r5f_pytest.txt
timeout -k 10 200 python tools/fuzz_q24.py 100 1 > gpurun_out/r5f_fuzz_q24.txt 2>&1; tail -2 gpurun_out/r5f_fuzz_q24.txt
timeout -k 10 200 python tools/fuzz_cc.py 100 5 > gpurun_out/r5f_fuzz_cc.txt 2>&1; tail -1 gpurun_out/r5f_fuzz_cc.txt
timeout -k 10 200 python tools/fuzz_state.py 100 5 > gpurun_out/r5f_fuzz_state.txt 2>&1; tail -1 gpurun_out/r5f_fuzz_state.txt
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > gpurun_out/r5f_bench.txt 2>&1; tail -1 gpurun_out/r5f_bench.txt | cut -c1-600
