# round 5 (second session): profiles of the final build -- stamps, rocprofv3 kernel trace + PMC (r05d), bench lines
timeout -k 10 300 python tools/stamps.py > gpurun_out/r5A_stamps_q24.txt 2>&1; tail -30 gpurun_out/r5A_stamps_q24.txt
tools/prof.sh r05d > gpurun_out/r5A_prof.txt 2>&1; tail -3 gpurun_out/r5A_prof.txt | cut -c1-300
timeout -k 10 400 python bench.py --steps 20 --warmup 2 > gpurun_out/r05d_bench_line.txt 2>&1; tail -1 gpurun_out/r05d_bench_line.txt | cut -c1-200
timeout -k 10 400 python bench.py --stack --days 30 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05d_bench_line_stack.txt 2>&1; tail -1 gpurun_out/r05d_bench_line_stack.txt | cut -c1-200
