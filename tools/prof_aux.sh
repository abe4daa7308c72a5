#!/bin/bash
# rocprofv3 kernel-trace summaries of the paths beside the headline bench (run on the GPU box):
#   post-CC chain (tools/time_post.py), any-length transforms (tools/time_fft.py), timelapse (tools/bench_timelapse.py)
# usage: tools/prof_aux.sh <tag>
set -o pipefail
tag=$1
root=$(pwd)
out=$root/gpurun_out/aux_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
python3 $root/tools/time_post.py 1024 > $out/time_post.txt 2>&1 || { tail -20 $out/time_post.txt; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $out/post -- python3 $root/tools/time_post.py 1024 > $out/time_post_traced.txt 2>&1 || { tail -20 $out/time_post_traced.txt; exit 1; }
python3 $root/tools/time_fft.py 128 > $out/time_fft.txt 2>&1 || { tail -20 $out/time_fft.txt; exit 1; }
python3 $root/tools/bench_timelapse.py --steps 3 > $out/timelapse.txt 2>&1 || { tail -20 $out/timelapse.txt; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tl -- python3 $root/tools/bench_timelapse.py --steps 3 --no-cpu-baseline > $out/timelapse_traced.txt 2>&1 || { tail -20 $out/timelapse_traced.txt; exit 1; }
cd $root
python3 bench.py --stations 16 --no-preprocess --steps 5 --warmup 1 > $out/bench_config2.txt 2>&1 || { tail -20 $out/bench_config2.txt; exit 1; }
find $out/post -name '*kernel_stats.csv' -exec cp {} $out/post_kernel_stats.csv \;
find $out/tl -name '*kernel_stats.csv' -exec cp {} $out/timelapse_kernel_stats.csv \;
rm -rf $out/post $out/tl
cat $out/time_post.txt $out/time_fft.txt; tail -1 $out/timelapse.txt; tail -1 $out/bench_config2.txt | cut -c1-600
