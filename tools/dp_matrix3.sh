#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for q in 2 3 4 5 6 8; do
  r=$(GPU_MAX_HW_QUEUES=$q NOAUX=1 MODE=full CONFIG=ctc timeout -k 10 120 python3 $R/tools/dp_probe.py 2>/dev/null | grep -E "^plain  |^DataParallel" | sed 's/  */ /g' | tr '\n' '|')
  echo "ctc side=low noaux GPU_MAX_HW_QUEUES=$q  $r"
done
