#!/bin/bash
# Run HERE after `gpurun -- tools/prof.sh <tag>`: copy the judged summaries from the scratch
# gpurun_out/prof_<tag>/ into profiles/ (tracked).   usage: tools/collect_prof.sh <tag>
set -e
tag=$1
out=gpurun_out/prof_$tag
mkdir -p profiles
cp $out/summary.txt profiles/${tag}_summary.txt
cp $out/summary.json profiles/${tag}_summary.json
cp $out/pmc_latest.json profiles/pmc_latest.json
find $out/kt -name '*kernel_stats.csv' -exec cp {} profiles/${tag}_kernel_stats.csv \;
grep -h '^{' $out/bench_kt.log > profiles/${tag}_bench_line_under_trace.json || true
echo "profiles/${tag}_* written"
