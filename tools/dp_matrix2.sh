#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ctc joint; do
for q in 2 3 4 5 6 8; do
  r=$(GPU_MAX_HW_QUEUES=$q SIDE=normal MODE=full CONFIG=$cfg timeout -k 10 120 python3 $R/tools/dp_probe.py 2>/dev/null | grep -E "^plain  |^DataParallel" | sed 's/  */ /g' | tr '\n' '|')
  echo "$cfg side=normal GPU_MAX_HW_QUEUES=$q  $r"
done; done
