#!/bin/bash
# ms/step (tools/step_time.py) under a list of runtime environment settings, one process each, two rounds on the same box:
# bash tools/rt_env_sweep.sh "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=8" ...      (CONFIG=joint for the joint model)
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for e in "BASE=1" "$@"; do
  # every setting under its own limit: a setting under which the step makes no progress (round 3: ROC_SYSTEM_SCOPE_SIGNAL=0) costs
  # PER_SETTING_S seconds, not the call; a timed-out setting ends the sweep (no further GPU step behind a hang)
  r=$(timeout -k 10 ${PER_SETTING_S:-90} env $e python3 $R/tools/step_time.py ${CONFIG:-} 2>/dev/null | grep "ms/step" | sed 's/.*min //')
  rc=${PIPESTATUS[0]}
  echo "round $round  $e  ->  $r"
  if [ "$rc" = "124" ] || [ "$rc" = "137" ]; then echo "setting $e hit its ${PER_SETTING_S:-90}-s limit: stopping the sweep"; exit 124; fi
done
done
