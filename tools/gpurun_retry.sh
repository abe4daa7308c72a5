#!/bin/bash
# gpurun with retries ONLY for "no box or slot free right now" (exit code 3: nothing ran, nothing charged).
# usage: tools/gpurun_retry.sh <timeout> '<command>'
t=$1; shift
for i in $(seq 20); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
