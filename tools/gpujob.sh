#!/bin/bash
# tools/gpujob.sh TIMEOUT 'command' -- gpurun with a wait for a free slot: exit code 3 (no box or slot free, nothing charged, nothing ran)
# is the only outcome that is tried again; whatever the command itself does is final.
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
