#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { r=$(env "$@" MODE=full timeout -k 10 120 python3 $R/tools/dp_probe.py 2>/dev/null | grep -E "^plain  |^DataParallel" | sed 's/  */ /g' | tr '\n' '|'); echo "$*  $r"; }
for q in 2 4 5 8; do
run MAINSTREAM=pool CONFIG=ctc GPU_MAX_HW_QUEUES=$q
run MAINSTREAM=pool CONFIG=joint GPU_MAX_HW_QUEUES=$q
done
