#!/bin/bash
# Instruction-mix counters of the solve sweeps (rocprofv3 PMC, separate passes).  Usage on the GPU box: bash tools/pmc_sweeps.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sweeps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-parity --steps 1 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- $BENCH > /dev/null 2> $OUT/p1.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -- $BENCH > /dev/null 2> $OUT/p2.err
cd $ROOT
python - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ddh::", "")[:48]
            if "solve_" not in k and "fft_axis_kernel<3, true" not in k and "gridwave" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(p, k, {c: "%.3g" % x for c, x in v.items()})
PY
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete
