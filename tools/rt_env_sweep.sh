#!/bin/bash
# ms/step (tools/step_time.py) under a list of runtime environment settings, one process each, two rounds on the same box:
# bash tools/rt_env_sweep.sh "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=8" ...      (CONFIG=joint for the joint model)
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for e in "BASE=1" "$@"; do
  r=$(env $e python3 $R/tools/step_time.py ${CONFIG:-} 2>/dev/null | grep "ms/step" | sed 's/.*min //')
  echo "round $round  $e  ->  $r"
done
done
