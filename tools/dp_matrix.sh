#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ctc joint; do
for side in low normal; do
for burn in 0 1 2 3; do
  r=$(SIDE=$side BURN=$burn MODE=full CONFIG=$cfg timeout -k 10 120 python3 $R/tools/dp_probe.py 2>/dev/null | grep -E "^plain  |^DataParallel" | sed 's/  */ /g' | tr '\n' '|')
  echo "$cfg side=$side burn=$burn  $r"
done; done; done
