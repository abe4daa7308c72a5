#!/bin/bash
# rocprofv3 summaries of the bench command (run on the GPU box via gpurun).
# usage: tools/prof.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > $out/bench_kt.log 2>&1 || { tail -20 $out/bench_kt.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing "$@" > $out/bench_pmc_fetch.log 2>&1 || { tail -20 $out/bench_pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing "$@" > $out/bench_pmc_write.log 2>&1 || { tail -20 $out/bench_pmc_write.log; exit 1; }
cd $root
python3 tools/prof_summarize.py $out
