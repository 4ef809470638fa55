#!/bin/bash
# SQ counters of the fused y stage alone (tools/bench_fused.py), per kernel variant, two separate PMC passes each.
# Usage on the GPU box: bash tools/gpu/fused_pmc.sh <outdir>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-gpurun_out/fused_pmc}
case $OUT in /*) ;; *) OUT=$ROOT/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for envs in "DDH_GW_V2=0" "DDH_GW_V2=1"; do
  i=$((i+1))
  env $envs FUSED_DERIV=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/v${i}a -- python $ROOT/tools/bench_fused.py > /dev/null 2> $OUT/v${i}a.err
  env $envs FUSED_DERIV=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/v${i}b -- python $ROOT/tools/bench_fused.py > /dev/null 2> $OUT/v${i}b.err
done
cd $ROOT
python - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
lines = []
for p in ("v1a", "v1b", "v2a", "v2b"):
    files = glob.glob("%s/%s/**/*counter_collection.csv" % (out, p), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ddh::", "")[:60]
            if "gridwave" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        n = max(len(cnt[k]), 1)
        lines.append("%s %s disp=%d  " % (p, k, n) + "  ".join("%s=%.4g" % (c, x / n) for c, x in sorted(v.items())))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
