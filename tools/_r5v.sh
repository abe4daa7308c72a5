# round 5 (second session): fuzzers on the zero-skip / nt-forward build (6 minutes each)
timeout -k 10 420 python tools/fuzz_q24.py 360 11 > gpurun_out/r5v_fuzz_q24.txt 2>&1; tail -1 gpurun_out/r5v_fuzz_q24.txt
timeout -k 10 420 python tools/fuzz_cc.py 360 12 > gpurun_out/r5v_fuzz_cc.txt 2>&1; tail -1 gpurun_out/r5v_fuzz_cc.txt
timeout -k 10 300 python tools/fuzz_state.py 240 13 > gpurun_out/r5v_fuzz_state.txt 2>&1; tail -1 gpurun_out/r5v_fuzz_state.txt
