#!/bin/bash
# Profiles of the benchmark on the GPU box (rocprofv3): kernel trace + stats, and three PMC passes
# (SQ_*, FETCH_SIZE, WRITE_SIZE collected separately, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage (from the repo root on the GPU box):  bash tools/profile_round.sh <tag>
# Writes gpurun_out/prof_<tag>/...; copy the summaries under profiles/ afterwards.
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline"
# 1. kernel trace + stats over the default-length run (same command line as the committed bench JSON)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --steps 5 --warmup 2 > $OUT/bench_stats_run.json 2> $OUT/stats.err
# 2..4. PMC passes on a short run (1 warm-up + 1 timed step)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_write.err
cd $ROOT
python tools/profile_summary.py $OUT $TAG
# raw CSVs are large: keep only the summaries and the stats table
find $OUT -name "*_counter_collection.csv" -delete
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
ls -la $OUT
