# round 5 (third session): configs[4] line of the final library with roofline.traffic from the PMC passes of the same build
timeout -k 10 200 python tools/bench_timelapse.py --steps 5 > gpurun_out/r05k_timelapse_bench_line2.txt 2>&1; tail -1 gpurun_out/r05k_timelapse_bench_line2.txt | cut -c1-900
