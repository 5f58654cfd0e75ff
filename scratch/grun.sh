#!/bin/bash
# usage: scratch/grun.sh LOGNAME TIMEOUT 'command'   -- gpurun with retries while no GPU slot is free (exit code 3: nothing charged)
log=gpurun_out/$1.log; to=$2; shift 2
for attempt in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "[grun] rc=$rc" >> $log; exit $rc; fi
  echo "[grun] no slot (attempt $attempt), retrying in 60 s" >> $log
  sleep 60
done
