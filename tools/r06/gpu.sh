#!/bin/bash
# container-side helper: gpurun with retries while the pool is busy.  usage: tools/r06/gpu.sh <timeout_s> <logfile> <command...>
T=$1; LOG=$2; shift 2
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $LOG; then exit $rc; fi
  sleep 60
done
exit 3
