timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5j_pytest.txt 2>&1
tail -4 gpurun_out/r5j_pytest.txt
tools/prof.sh r05b > gpurun_out/r5j_prof.txt 2>&1; tail -3 gpurun_out/r5j_prof.txt | cut -c1-300
