#!/bin/bash
# ms/step of the plain step and of the one-rank RCCL data-parallel step under GPU_MAX_HW_QUEUES = 1..8 (the HIP runtime reads it when it
# starts: one process per value).  bash tools/dp_queue_sweep.sh [joint]
R=${GRAFT_REPO_ROOT:-/root/repo}
for q in 1 2 3 4 5 6 8; do
  r=$(GPU_MAX_HW_QUEUES=$q MODE=full CONFIG=${1:-ctc} timeout -k 10 120 python3 $R/tools/dp_probe.py 2>/dev/null | grep -E "^plain  |^DataParallel" | sed 's/  */ /g' | tr '\n' '|')
  echo "GPU_MAX_HW_QUEUES=$q  $r"
done
