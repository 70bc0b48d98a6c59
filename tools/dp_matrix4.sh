#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { r=$(env "$@" MODE=full CONFIG=ctc timeout -k 10 120 python3 $R/tools/dp_probe.py 2>/dev/null | grep -E "^plain  |^DataParallel" | sed 's/  */ /g' | tr '\n' '|'); echo "ctc $*  $r"; }
run ASR_ARMED_FORK=0
run TORCH_NCCL_HIGH_PRIORITY=1
run ASR_WGRAD_OVERLAP=0
run NCCL_MAX_NCHANNELS=4
run HIP_FORCE_DEV_KERNARG=0
