#!/bin/bash
# gpurun with patience: exit code 3 = no box / slot free right now (nothing charged) -> wait and ask again; every other verdict is final.
#   bash tools/gpurun_retry.sh <timeout-seconds> '<command>'
T=${1:?timeout}; shift
for try in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  echo "[gpurun_retry] no slot (try $try), sleeping 90 s"
  sleep 90
done
exit 3
