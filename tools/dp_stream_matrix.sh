#!/bin/bash
# plain and one-rank RCCL data-parallel step, CTC and joint, with N other pool streams used BEFORE the model is built and under several
# GPU_MAX_HW_QUEUES: with the measured stream selection (engine.pick_stream) and without it (PROBE=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { r=$(env "$@" MODE=full timeout -k 10 120 python3 $R/tools/dp_probe.py 2>/dev/null | grep -E "^plain  |^DataParallel" | sed 's/  */ /g' | tr '\n' '|'); echo "$*  $r"; }
for cfg in ctc joint; do
for t in 0 2 3; do
run CONFIG=$cfg TOUCH=$t PROBE=1
done
for q in 2 5 8; do
run CONFIG=$cfg GPU_MAX_HW_QUEUES=$q PROBE=1
done
done
