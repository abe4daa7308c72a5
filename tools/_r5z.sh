# round 5 (second session): scalar descriptor loads + branch-free row-wave loads -- suite + A/B
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5z_pytest.txt 2>&1; tail -3 gpurun_out/r5z_pytest.txt
tools/ab.sh "libxcx_prev.so libxcx.so" 4 > gpurun_out/r5z_ab.txt 2>&1
cat gpurun_out/r5z_ab.txt
