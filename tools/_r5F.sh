# round 5 (second session): final suite + profiles of the final build (r05e)
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5F_pytest.txt 2>&1; tail -3 gpurun_out/r5F_pytest.txt
tools/prof.sh r05e > gpurun_out/r5F_prof.txt 2>&1; tail -3 gpurun_out/r5F_prof.txt | cut -c1-300
timeout -k 10 300 python tools/stamps.py > gpurun_out/r5F_stamps_q24.txt 2>&1; tail -28 gpurun_out/r5F_stamps_q24.txt | head -12
