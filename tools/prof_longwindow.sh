#!/bin/bash
# Production-window paths on the GPU box: tools/time_longwindow.py (one pair-day, 8 pair-days, all pairs of 16 / 64 stations),
# tools/time_post_long.py (post-CC chain on 1.62 M-lag rows), and the kernel trace of the 64-station case.
# usage: tools/prof_longwindow.sh <tag>      (then copy gpurun_out/lw_<tag>/* into profiles/ as <tag>_longwindow_*)
set -o pipefail
tag=$1
root=$(pwd)
out=$root/gpurun_out/lw_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
python3 $root/tools/time_longwindow.py 1 16 > $out/times.txt 2>&1 || { tail -20 $out/times.txt; exit 1; }
python3 $root/tools/time_longwindow.py 8 64 >> $out/times.txt 2>&1 || { tail -20 $out/times.txt; exit 1; }
python3 $root/tools/time_post_long.py 64 >> $out/times.txt 2>&1 || { tail -20 $out/times.txt; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o lw -- python3 $root/tools/time_longwindow.py 8 64 > $out/traced.txt 2>&1 || { tail -20 $out/traced.txt; exit 1; }
cp $out/kt/lw_kernel_stats.csv $out/kernel_stats.csv && rm -rf $out/kt
cat $out/times.txt
