timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5k_pytest.txt 2>&1
tail -3 gpurun_out/r5k_pytest.txt
timeout -k 10 400 python bench.py --gpus 4 --oversubscribe --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05_selflaunch_n4.txt 2>&1; echo "rc=$?" >> gpurun_out/r05_selflaunch_n4.txt; tail -2 gpurun_out/r05_selflaunch_n4.txt | cut -c1-500
tools/prof.sh r05c > gpurun_out/r5k_prof.txt 2>&1; tail -3 gpurun_out/r5k_prof.txt | cut -c1-300
timeout -k 10 400 python bench.py --steps 20 --warmup 2 > gpurun_out/r05_bench_line.txt 2>&1; tail -1 gpurun_out/r05_bench_line.txt | cut -c1-200
