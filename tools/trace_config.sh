#!/bin/bash
# kernel trace of one bench configuration on the GPU box: bash tools/trace_config.sh <tag> <bench args...>
# -> gpurun_out/<tag>_stats.txt (per-step kernel totals), gpurun_out/<tag>_timeline.txt (one step, every kernel with start / duration / queue)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o p --output-format csv -- python3 $R/bench.py "$@" --steps 12 --warmup 6 --no-extras --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_trace.err
cd $R
CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/trace_stats.py $CSV > gpurun_out/${TAG}_stats.txt
python3 tools/timeline.py $CSV 8 9 dump > gpurun_out/${TAG}_timeline.txt
rm -rf $OUT
tail -3 gpurun_out/${TAG}_stats.txt
