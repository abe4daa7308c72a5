# round 5 (second session): fuzzers on the FINAL library (4 + 4 + 3 minutes)
timeout -k 10 300 python tools/fuzz_q24.py 240 21 > gpurun_out/r5N_fuzz_q24.txt 2>&1; tail -1 gpurun_out/r5N_fuzz_q24.txt
timeout -k 10 300 python tools/fuzz_cc.py 240 22 > gpurun_out/r5N_fuzz_cc.txt 2>&1; tail -1 gpurun_out/r5N_fuzz_cc.txt
timeout -k 10 240 python tools/fuzz_state.py 180 23 > gpurun_out/r5N_fuzz_state.txt 2>&1; tail -1 gpurun_out/r5N_fuzz_state.txt
timeout -k 10 240 python tools/fuzz_ops.py 150 24 > gpurun_out/r5N_fuzz_ops.txt 2>&1; tail -1 gpurun_out/r5N_fuzz_ops.txt | cut -c1-600
