#!/bin/bash
# usage: grun.sh <timeout_s> '<command>' — retries while no GPU slot is free (gpurun rc 3)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/grun_last.log 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/grun_last.log; then sleep 90; continue; fi
  break
done
tail -40 /tmp/grun_last.log
exit $rc
