#!/bin/bash
# PMC passes over the shell band LU at config H (tools/gpu/ellband_time.py): SQ wait / issue / active fractions and HBM
# traffic per kernel, separate passes as MI355X_MICROARCH.md prescribes.  Usage on the GPU box: bash tools/gpu/ellband_pmc.sh <tag>
set -u
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_eb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/gpu/ellband_time.py"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > /dev/null 2> $OUT/pmc_sq.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.err
cd $ROOT
python - $OUT <<'PY'
import os, sys
sys.path.insert(0, "tools")
from profile_summary import read_counters
out = sys.argv[1]
sq, fe, wr = (read_counters(os.path.join(out, d)) for d in ("pmc_sq", "pmc_fetch", "pmc_write"))
lines = ["# rocprofv3 --kernel-trace --pmc (separate passes: SQ_*, FETCH_SIZE, WRITE_SIZE) -- python tools/gpu/ellband_time.py",
         "# MI355X, shell convection ShellBasis(256,128,128): 21 solves + 7 factorizations.  HBM read = 2 x FETCH_SIZE KiB (gfx950), write = WRITE_SIZE KiB",
         "%-44s %4s %10s %10s | %8s %8s %8s %8s" % ("kernel", "n", "rd_GB", "wr_GB", "wait_any", "wait_ins", "active", "ldsconf")]
for k in sorted(set(sq) | set(fe) | set(wr)):
    if "ellband" not in k:
        continue
    s = sq.get(k, {})
    wc = sum(s.get("SQ_WAVE_CYCLES", [])) or float("nan")
    la = sum(s.get("SQ_LDS_IDX_ACTIVE", []))
    mean = lambda a: sum(a) / len(a) if a else float("nan")
    rd = [2.0 * 1024.0 * v / 1e9 for v in fe.get(k, {}).get("FETCH_SIZE", [])]
    ww = [1024.0 * v / 1e9 for v in wr.get(k, {}).get("WRITE_SIZE", [])]
    lines.append("%-44s %4d %10.3f %10.3f | %8.2f %8.2f %8.2f %8.2f" % (
        k[:44], len(s.get("SQ_WAVE_CYCLES", [])), mean(rd), mean(ww), sum(s.get("SQ_WAIT_ANY", [])) / wc,
        sum(s.get("SQ_WAIT_INST_ANY", [])) / wc, sum(s.get("SQ_ACTIVE_INST_ANY", [])) / wc,
        (sum(s.get("SQ_LDS_BANK_CONFLICT", [])) / la) if la else 0.0))
open(os.path.join(out, "ellband_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT -name "*_counter_collection.csv" -delete
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
