#!/bin/bash
# interleaved A/B of library builds on one GPU box: tools/ab.sh "libA libB ..." rounds [bench args]
libs=$1; rounds=$2; shift; shift
for r in $(seq $rounds); do for l in $libs; do
  XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), round(d['ms_per_step'],1))"
done; done
