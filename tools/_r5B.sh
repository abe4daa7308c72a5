# round 5 (second session): second row item through LDS-DMA -- suite + A/B
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5B_pytest.txt 2>&1; tail -3 gpurun_out/r5B_pytest.txt
tools/ab.sh "libxcx_nodma.so libxcx.so" 4 > gpurun_out/r5B_ab.txt 2>&1
cat gpurun_out/r5B_ab.txt
