#!/bin/bash
# developer helper: gpurun with retries while no GPU slot / box is free (exit code 3: nothing charged).  usage: tools/gpurun_retry.sh <timeout_s> '<command>'
T=$1; shift
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[retry] attempt $attempt: no slot, sleeping 45 s" >&2
  sleep 45
done
exit 3
