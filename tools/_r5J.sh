# round 5 (second session): final suite + profiles + bench lines of the final build (r05i)
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5J_pytest.txt 2>&1; tail -3 gpurun_out/r5J_pytest.txt
tools/prof.sh r05i > gpurun_out/r5J_prof.txt 2>&1; tail -3 gpurun_out/r5J_prof.txt | cut -c1-200
timeout -k 10 300 python tools/stamps.py > gpurun_out/r5J_stamps_q24.txt 2>&1; tail -28 gpurun_out/r5J_stamps_q24.txt | grep -E 'lifetime|launch ->|staging|LDS exchange'
