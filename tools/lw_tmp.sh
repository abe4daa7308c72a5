root=$(pwd); out=$root/gpurun_out/r4lw; mkdir -p $out
python3 tools/time_longwindow.py 1; python3 tools/time_longwindow.py 8
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -- python3 $root/tools/time_longwindow.py 8 > $out/log.txt 2>&1
find $out/p -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \; ; rm -rf $out/p
head -12 $out/kernel_stats.csv | cut -c1-100,240-330
