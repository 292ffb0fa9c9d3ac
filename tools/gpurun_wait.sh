#!/bin/bash
# gpurun, waiting for a free slot: repeats the call ONLY while gpurun answers "no box or slot free right now" (exit code 3, nothing ran, nothing
# charged); any other outcome -- success, a failing command, a refusal -- is returned as it is.   tools/gpurun_wait.sh [--timeout S] -- '<command>'
for attempt in $(seq 1 30); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
