#!/bin/bash
# wall ms/step of the CTC config (tools/host_overhead.py) under a list of environment settings, same box, two rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for e in "BASE=1" "$@"; do
  r=$(env $e python3 $R/tools/host_overhead.py ${CONFIG:-} 2>/dev/null | grep "host enqueue" | sed 's/.*wall //')
  echo "round $round  $e  ->  $r"
done
done
