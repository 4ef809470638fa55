#!/bin/bash
# Dense-contraction evidence of the secondary configurations (rocprofv3, separate passes):
#   H (shell): FP64 MFMA instruction / flop counters and kernel times of the SWSH GEMM and the ell-term GEMM;
#   S (sphere): HBM traffic (FETCH_SIZE, WRITE_SIZE) and kernel times of the grouped GEMV and the per-m batched GEMV.
# Usage on the GPU box: bash tools/gpu/mfma_pmc.sh <outdir>   -> <outdir>/shell_mfma.txt, <outdir>/sphere_gemv.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-gpurun_out/mfma_pmc}
case $OUT in /*) ;; *) OUT=$ROOT/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
H="python $ROOT/tools/bench_configs.py shell"
S="python $ROOT/tools/bench_configs.py sphere"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h_stats -- $H > $OUT/h_stats.log 2> $OUT/h_stats.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/h_pmc -- $H > /dev/null 2> $OUT/h_pmc.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_stats -- $S > $OUT/s_stats.log 2> $OUT/s_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/s_fetch -- $S > /dev/null 2> $OUT/s_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/s_write -- $S > /dev/null 2> $OUT/s_write.err
cd $ROOT
python - $OUT <<'PY'
import csv, glob, collections, re, sys
out = sys.argv[1]
def short(n):
    n = re.sub(r"\(.*$", "", n).strip(); n = re.sub(r"^void\s+", "", n); return n.replace("ddh::", "")[:58]
def stats(d):
    t = {}
    for f in glob.glob("%s/%s/**/*kernel_stats.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(f)):
            t[short(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    return t
def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}
# ---- H: FP64 MFMA
ht = stats("h_stats"); hc, hn = counters("h_pmc")
L = ["# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python tools/bench_configs.py shell  (H shell convection ShellBasis(256, 128, 128) SBDF2; round-5 build)",
     "# kernels that issue FP64 MFMA instructions; flop = MOPS_F64 x 512; time = total duration of the same kernel in a --kernel-trace --stats run of the same command; peak FP64 matrix = 78.6 TFLOP/s",
     "%-58s %6s %14s %14s %10s %9s %8s" % ("kernel", "disp", "mfma_f64_inst", "MOPS_F64", "time_ms", "TFLOP/s", "of peak")]
for k, v in sorted(hc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)):
    if v.get("SQ_INSTS_VALU_MFMA_F64", 0) <= 0: continue
    ms = ht.get(k, (0, float("nan")))[1]
    # the PMC run and the stats run execute the same command: scale the time to the dispatches counted
    calls = ht.get(k, (hn[k], 0))[0]
    flops = v["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 * (calls / hn[k])
    tf = flops / (ms * 1e-3) / 1e12 if ms == ms and ms > 0 else float("nan")
    L.append("%-58s %6d %14.0f %14.0f %10.3f %9.2f %8.3f" % (k, hn[k], v["SQ_INSTS_VALU_MFMA_F64"], v["SQ_INSTS_VALU_MFMA_MOPS_F64"], ms, tf, tf / 78.6))
open(out + "/shell_mfma.txt", "w").write("\n".join(L) + "\n"); print("\n".join(L))
# ---- S: traffic of the GEMV kernels
st = stats("s_stats"); fe, fn = counters("s_fetch"); wr, wn = counters("s_write")
L = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/bench_configs.py sphere  (S shallow water SphereBasis(512, 256) RK222; round-5 build)",
     "# HBM read bytes = 2 x FETCH_SIZE x 1024 on gfx950 (MI355X_MICROARCH.md), WRITE_SIZE x 1024 as is; per-launch means; time from a --stats run of the same command",
     "%-58s %6s %12s %12s %10s %9s" % ("kernel", "calls", "rd_MB/launch", "wr_MB/launch", "us/launch", "TB/s")]
for k in sorted(set(fe) | set(wr), key=lambda k: -st.get(k, (0, 0))[1]):
    if k not in st or st[k][1] < 0.5: continue
    rd = 2 * 1024 * fe.get(k, {}).get("FETCH_SIZE", 0.0) / max(fn.get(k, 1), 1)
    ww = 1024 * wr.get(k, {}).get("WRITE_SIZE", 0.0) / max(wn.get(k, 1), 1)
    us = st[k][1] * 1e3 / st[k][0]
    L.append("%-58s %6d %12.2f %12.2f %10.2f %9.2f" % (k, st[k][0], rd / 1e6, ww / 1e6, us, (rd + ww) / (us * 1e-6) / 1e12))
open(out + "/sphere_gemv.txt", "w").write("\n".join(L) + "\n"); print("\n".join(L))
PY
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*_stats.csv" -delete
