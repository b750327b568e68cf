#!/bin/bash
# Run GPU steps one after the other on a gpurun box: each line of stdin is "<seconds> <logfile> <command...>".  A step that is
# KILLED at its limit (timeout: 124 / 137) ends the whole call -- no further GPU step after a hang; a step that merely fails
# (assertion, non-zero exit) is logged and the next one runs.
mkdir -p gpurun_out/r4
while read -r secs log cmd; do
    [ -z "$secs" ] && continue
    echo "=== [$secs s] $cmd > $log"
    timeout -k 10 "$secs" bash -c "$cmd" > "$log" 2>&1
    rc=$?
    echo "rc=$rc" >> "$log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "step killed at its limit (rc=$rc): stopping"; tail -5 "$log"; exit $rc
    fi
    tail -n "${TAILN:-6}" "$log"
done
