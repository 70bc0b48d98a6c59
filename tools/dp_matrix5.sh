#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { r=$(env "$@" MODE=full CONFIG=ctc timeout -k 10 120 python3 $R/tools/dp_probe.py 2>/dev/null | grep -E "^plain  |^DataParallel" | sed 's/  */ /g' | tr '\n' '|'); echo "ctc $*  $r"; }
run TORCH_NCCL_ASYNC_ERROR_HANDLING=0
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_DUMP_ON_TIMEOUT=0
run GPU_MAX_HW_QUEUES=4 DEBUG_CLR_LIMIT_BLIT_WG=0
