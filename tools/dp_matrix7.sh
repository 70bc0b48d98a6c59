#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { r=$(env "$@" MODE=full timeout -k 10 120 python3 $R/tools/dp_probe.py 2>/dev/null | grep -E "^plain  |^DataParallel" | sed 's/  */ /g' | tr '\n' '|'); echo "$*  $r"; }
for t in 1 2 3 4; do
run TOUCH=$t CONFIG=ctc
run TOUCH=$t CONFIG=joint
done
