# round 5 (second session): bench lines of the final build
timeout -k 10 400 python bench.py --steps 20 --warmup 2 > gpurun_out/r05i_bench_line.txt 2>&1; tail -1 gpurun_out/r05i_bench_line.txt | cut -c1-200
timeout -k 10 400 python bench.py --stack --days 30 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05i_bench_line_stack.txt 2>&1; tail -1 gpurun_out/r05i_bench_line_stack.txt | cut -c1-200
timeout -k 10 300 python bench.py --stations 16 --no-preprocess --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r05i_config2_bench_line.txt 2>&1; tail -1 gpurun_out/r05i_config2_bench_line.txt | cut -c1-200
timeout -k 10 300 python tools/bench_timelapse.py --steps 3 --no-cpu-baseline > gpurun_out/r05i_timelapse_bench_line.txt 2>&1; tail -1 gpurun_out/r05i_timelapse_bench_line.txt | cut -c1-200
timeout -k 10 400 python bench.py --gpus 2 --oversubscribe --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05i_selflaunch_n2.txt 2>&1; tail -1 gpurun_out/r05i_selflaunch_n2.txt | cut -c1-200
