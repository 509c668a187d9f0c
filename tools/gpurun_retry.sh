#!/bin/bash
# gpurun with retries while no GPU slot / box is free (exit code 3: nothing charged).  Usage: tools/gpurun_retry.sh <timeout s> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
