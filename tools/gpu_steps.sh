#!/bin/bash
# Runs GPU steps one after the other on a gpurun box: each line of the step file is `SECONDS LOGNAME COMMAND...`.
# A step that fails in the ordinary way (non-zero exit) does not stop the next one; a step that is killed by its time limit
# (exit 124 / 137) ends the whole call - no further GPU step is started behind a hang.
# usage: tools/gpu_steps.sh steps.txt
mkdir -p gpurun_out
while IFS= read -r line; do
    [ -z "$line" ] && continue
    secs=${line%% *}; rest=${line#* }; name=${rest%% *}; cmd=${rest#* }
    echo "== step $name (limit ${secs}s): $cmd"
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
    rc=$?
    echo "== step $name rc=$rc"
    tail -n 3 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "== step $name hit its time limit: stopping"
        exit $rc
    fi
done < "$1"
exit 0
