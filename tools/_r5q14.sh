# round 5 (third session): final suite + profiles + bench lines of the final build (r05k)
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5q14_pytest.txt 2>&1; rc=$?; tail -3 gpurun_out/r5q14_pytest.txt
[ $rc -ne 0 ] && exit 1
tools/prof.sh r05k > gpurun_out/r5q14_prof.txt 2>&1 || { tail -20 gpurun_out/r5q14_prof.txt; exit 1; }
tail -3 gpurun_out/r5q14_prof.txt | cut -c1-200
cp gpurun_out/prof_r05k/pmc_latest.json profiles/pmc_latest.json       # the bench lines below carry roofline.traffic of THIS build
timeout -k 10 400 python bench.py --steps 20 --warmup 2 > gpurun_out/r05k_bench_line.txt 2>&1 || exit 1; tail -1 gpurun_out/r05k_bench_line.txt | cut -c1-200
timeout -k 10 400 python bench.py --stack --days 30 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05k_bench_line_stack.txt 2>&1 || exit 1; tail -1 gpurun_out/r05k_bench_line_stack.txt | cut -c1-200
timeout -k 10 300 python bench.py --stations 16 --no-preprocess --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r05k_config2_bench_line.txt 2>&1 || exit 1; tail -1 gpurun_out/r05k_config2_bench_line.txt | cut -c1-200
timeout -k 10 400 python tools/bench_timelapse.py --steps 5 > gpurun_out/r05k_timelapse_bench_line.txt 2>&1 || exit 1; tail -1 gpurun_out/r05k_timelapse_bench_line.txt | cut -c1-300
timeout -k 10 400 python bench.py --gpus 2 --oversubscribe --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05k_selflaunch_n2.txt 2>&1 || exit 1; tail -1 gpurun_out/r05k_selflaunch_n2.txt | cut -c1-200
XCX_CC2D_OVERLAP=0 tools/pmc_cc2d.sh r05k_seq > gpurun_out/r5q14_cc2d_seq.txt 2>&1; tail -8 gpurun_out/r5q14_cc2d_seq.txt | cut -c1-60,180-270
XCX_CC2D_OVERLAP=1 tools/pmc_cc2d.sh r05k > gpurun_out/r5q14_cc2d.txt 2>&1; tail -8 gpurun_out/r5q14_cc2d.txt | cut -c1-60,180-270
