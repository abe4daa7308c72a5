# round 5 (third session): fuzzers on the FINAL library (r05k): 2 + 2 + 2 + 2.5 + 2 minutes
timeout -k 10 200 python tools/fuzz_q24.py 120 31 > gpurun_out/r5q15_fuzz_q24.txt 2>&1; tail -1 gpurun_out/r5q15_fuzz_q24.txt
timeout -k 10 200 python tools/fuzz_cc.py 120 32 > gpurun_out/r5q15_fuzz_cc.txt 2>&1; tail -1 gpurun_out/r5q15_fuzz_cc.txt
timeout -k 10 200 python tools/fuzz_state.py 120 33 > gpurun_out/r5q15_fuzz_state.txt 2>&1; tail -1 gpurun_out/r5q15_fuzz_state.txt
timeout -k 10 240 python tools/fuzz_ops.py 150 34 > gpurun_out/r5q15_fuzz_ops.txt 2>&1; tail -1 gpurun_out/r5q15_fuzz_ops.txt | cut -c1-700
timeout -k 10 200 python tools/check_cc2d.py 120 35 > gpurun_out/r5q15_check_cc2d.txt 2>&1; tail -1 gpurun_out/r5q15_check_cc2d.txt
