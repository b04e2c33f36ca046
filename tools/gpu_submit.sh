#!/bin/bash
# tools/gpu_submit.sh LOG TIMEOUT CMD...: gpurun with a patient wait for a free box (exit 3 = nothing free, nothing charged: wait and
# ask again; any other exit code is the command's own and is final -- a failed GPU command is never re-run from here)
LOG=$1; shift; TMO=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
