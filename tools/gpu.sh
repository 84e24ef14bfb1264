#!/bin/bash
# tools/gpu.sh TIMEOUT 'command'  -- gpurun with retries while no box / slot is free (exit 3 = nothing charged)
T=$1; shift
for i in 1 2 3 4 5 6 7 8; do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    echo "[gpu.sh] no slot (try $i), sleeping 90 s"; sleep 90
done
exit 3
