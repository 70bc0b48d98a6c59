#!/bin/bash
# Round profiles of the headline bench command (run on the GPU box):  bash tools/profile_round.sh <tag>
# kernel trace + stats, then separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) as MI355X_MICROARCH.md prescribes.
# Writes the summaries under gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/.
set -e
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 10 --no-extras --no-cpu-baseline"      # the headline workload (round 5: the joint model, configs[2])
rocprofv3 --kernel-trace --stats -d $OUT/trace -o p --output-format csv -- python3 $R/bench.py $ARGS > $OUT/bench_profiled.json 2> $OUT/trace.err
echo "trace done"
# counter passes: on the CTC-only configuration (the same GEMM / attention / LayerNorm kernels at the same shapes; under counter collection the joint
# step's multi-queue hand-overs ended in HSA_STATUS_ERROR_INVALID_PACKET_FORMAT on this stack, round 5), few steps (kernels run one at a time in
# counter mode), each pass under its own time limit
PARGS="--config ctc --steps 10 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 $R/bench.py $PARGS > /dev/null 2> $OUT/fetch.err
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 $R/bench.py $PARGS > /dev/null 2> $OUT/write.err
echo "write done"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma -o p --output-format csv -- python3 $R/bench.py $PARGS > /dev/null 2> $OUT/mfma.err
echo "mfma done"
cd $R
python3 tools/trace_stats.py $(find $OUT/trace -name "*kernel_trace.csv") $OUT/kernel_stats_per_step.csv > $OUT/kernel_stats_per_step.txt
cp $(find $OUT/trace -name "*kernel_stats.csv") $OUT/kernel_stats.csv
python3 tools/pmc_summary.py $(find $OUT/fetch -name "*counter_collection.csv") $(find $OUT/write -name "*counter_collection.csv") $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
python3 tools/mfma_util.py $(find $OUT/mfma -name "*counter_collection.csv") $OUT/mfma_util.json
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/mfma
# the same for the CTC-only configuration (configs[1]): kernel trace only
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_ctc -o p --output-format csv -- python3 $R/bench.py $ARGS --config ctc > $OUT/bench_profiled_ctc.json 2> $OUT/trace_ctc.err
cd $R
python3 tools/trace_stats.py $(find $OUT/trace_ctc -name "*kernel_trace.csv") $OUT/ctc_kernel_stats_per_step.csv > $OUT/ctc_kernel_stats_per_step.txt
rm -rf $OUT/trace_ctc
echo "ctc trace done"
# HBM traffic of the long-form band attention kernels (BASELINE configs[4] per GPU: B = 8, T = 2000, +-50 frames), stand-alone
cd /tmp
B=8 T=2000 WINDOW=50 REPS=5 rocprofv3 --pmc FETCH_SIZE -d $OUT/bfetch -o p --output-format csv -- python3 $R/tools/sdpa_bench.py > $OUT/band_bench.txt 2> $OUT/bfetch.err
B=8 T=2000 WINDOW=50 REPS=5 rocprofv3 --pmc WRITE_SIZE -d $OUT/bwrite -o p --output-format csv -- python3 $R/tools/sdpa_bench.py > /dev/null 2> $OUT/bwrite.err
cd $R
python3 tools/pmc_summary.py $(find $OUT/bfetch -name "*counter_collection.csv") $(find $OUT/bwrite -name "*counter_collection.csv") $OUT/pmc_traffic_band.json > $OUT/pmc_traffic_band.txt
B=8 T=2000 WINDOW=50 python3 tools/sdpa_bench.py > $OUT/band_bench_plain.txt 2>&1
rm -rf $OUT/bfetch $OUT/bwrite
# SQ counters of the attention kernels at the config-2 shape (issue mix, LDS bank conflicts), two passes
cd /tmp
REPS=5 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -d $OUT/sq1 -o p --output-format csv -- python3 $R/tools/sdpa_bench.py > /dev/null 2> $OUT/sq1.err
REPS=5 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/sq2 -o p --output-format csv -- python3 $R/tools/sdpa_bench.py > /dev/null 2> $OUT/sq2.err
cd $R
python3 tools/sq_counters.py $OUT/sq_counters_attention.json $(find $OUT/sq1 $OUT/sq2 -name "*counter_collection.csv") > $OUT/sq_counters_attention.txt
rm -rf $OUT/sq1 $OUT/sq2
ls -la $OUT
