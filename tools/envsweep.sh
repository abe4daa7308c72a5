#!/bin/bash
# interleaved sweep of environment / argument variants on one GPU box
# usage: tools/envsweep.sh rounds "ENV=.. ENV=.. -- bench args" "..." ...
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    envs=${v%%--*}; args=${v#*--}
    [ "$envs" == "$v" ] && { envs=$v; args=""; }
    res=$(env $envs timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing $args 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config']['pair_batch'])" 2>&1)
    echo "[$v] $res"
  done
done
