#!/bin/bash
# usage: tools/gpu_retry.sh <timeout-seconds> <command string>   (ONE argument after the timeout: it is handed to gpurun as a single shell
# command line, e.g. tools/gpu_retry.sh 600 'python -m pytest tests -m gpu -x -q') -- retries while gpurun reports "no box / slot free" (rc 3)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
