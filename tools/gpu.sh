#!/bin/bash
# gpurun with retries while no GPU slot is free (rc 3): tools/gpu.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
