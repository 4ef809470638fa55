#!/bin/bash
# Round-3 evidence in one GPU call (from the repo root on the GPU box):  bash tools/profile_round3.sh
#   1. tools/profile_round.sh r3: rocprofv3 --kernel-trace --stats of the bench command + three PMC passes
#   2. secondary configurations (K, R2, S, H): steps/s, refactorization cost
#   3. rocprofv3 --kernel-trace --stats of S and H (correctly labelled this time)
#   4. FP64 MFMA counters of S and H: instruction counts x the instruction shape (v_mfma_f64_16x16x4: 2048 flop per wave
#      instruction) and the matrix pipe's busy cycles -> TFLOP/s from counters, not hand counts
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r3
mkdir -p $OUT
bash $ROOT/tools/profile_round.sh r3 > $OUT/profile_round.log 2>&1
cd $ROOT
python tools/bench_configs.py > $OUT/secondary.txt 2>&1
python tools/bench_configs.py sphere >> $OUT/secondary.txt 2>&1
python tools/bench_configs.py shell >> $OUT/secondary.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "mfma" | head -40 > $OUT/mfma_counters_avail.txt
for cfg in sphere shell; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -- python $ROOT/tools/bench_configs.py $cfg > $OUT/stats_$cfg.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_mfma_$cfg -- python $ROOT/tools/bench_configs.py $cfg > $OUT/pmc_mfma_$cfg.log 2>&1
done
cd $ROOT
python tools/profile_summary3.py $OUT
find $OUT -name "*_counter_collection.csv" -delete
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
ls -la $OUT
