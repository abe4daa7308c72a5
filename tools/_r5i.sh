timeout -k 10 400 python bench.py --steps 20 --warmup 2 > gpurun_out/r05_bench_line.txt 2>&1; tail -1 gpurun_out/r05_bench_line.txt | cut -c1-300
timeout -k 10 600 python bench.py --stack --days 30 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_line_stack.txt 2>&1; tail -1 gpurun_out/r05_bench_line_stack.txt | cut -c1-400
timeout -k 10 300 python bench.py --stations 16 --no-preprocess --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r05_config2_bench_line.txt 2>&1; tail -1 gpurun_out/r05_config2_bench_line.txt | cut -c1-300
timeout -k 10 400 python tools/bench_timelapse.py > gpurun_out/r05_timelapse_bench_line.txt 2>&1; tail -1 gpurun_out/r05_timelapse_bench_line.txt | cut -c1-700
( timeout -k 10 200 python tools/fuzz_cc.py 120 11 | tail -1; timeout -k 10 200 python tools/fuzz_state.py 120 11 | tail -1; timeout -k 10 200 python tools/fuzz_q24.py 120 11 | tail -1; timeout -k 10 300 python tools/fuzz_ops.py 120 11 | tail -1; timeout -k 10 200 python tools/fuzz_seg.py 90 11 | tail -1 ) > gpurun_out/r05_fuzz_summary.txt 2>&1
cat gpurun_out/r05_fuzz_summary.txt | cut -c1-400
