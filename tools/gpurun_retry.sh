#!/bin/bash
# build-container helper: gpurun with retries while no box / slot is free (exit code 3): tools/gpurun_retry.sh <timeout_s> '<command>'
t=$1; shift
for k in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
