#!/bin/bash
# instruction-mix and wait counters of the step's kernels (rocprofv3 PMC, three separate passes; SQ block only -- a pass
# with TCP_* counters aborted rocprofv3 on this pool, tools/gpu/r3l.sh)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-parity --steps 1 --warmup 1"
run() { tag=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -- $BENCH > /dev/null 2> $OUT/$tag.err; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run b SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS
run c SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
cd $ROOT
python - <<PY > $OUT/r3_sq_counters.txt
import csv, glob, collections, os
print("# rocprofv3 --kernel-trace --pmc <SQ counters> -- python bench.py --steps 1 --warmup 1 (3 separate passes a / b / c), MI355X, 3-D RB 512x512x256")
print("# per-dispatch means of the step's kernels; SQ_*_CYCLES / ACTIVE / WAIT are wave-cycle sums in the counter's own unit")
keep = ("solve_", "gridwave", "wave_rfft", "wave_cheb", "band_matvec")
for tag in "abc":
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ddh::", "")[:52]
            if not any(s in k for s in keep): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in sorted(acc):
        print("pass %s  %-52s disp=%-3d %s" % (tag, k, len(n[k]), "  ".join("%s=%.4g" % (c, x / len(n[k])) for c, x in sorted(acc[k].items()))))
PY
cat $OUT/r3_sq_counters.txt | head -40
find $OUT -name "*.csv" -delete
