"""Condense the secondary-configuration profiles of tools/profile_round3.sh: per-kernel rocprofv3 stats of S (sphere) and
H (shell), and FP64 matrix-core use from the PMC counters (instruction counts x instruction shape, busy cycles)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name).strip()
    name = re.sub(r"^void\s+", "", name)
    return name.replace("ddh::", "")


def main():
    out = sys.argv[1]
    for cfg, label in (("sphere", "S  sphere shallow water SphereBasis(512, 256) RK222"),
                       ("shell", "H  shell convection ShellBasis(256, 128, 128) SBDF2")):
        lines = ["# rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py %s   (%s; setup + timed steps + 3 "
                 "refactorizations)" % (cfg, label),
                 "%-72s %7s %12s %12s %7s" % ("kernel", "calls", "avg_ms", "total_ms", "%")]
        durs = {}
        for f in glob.glob(os.path.join(out, "stats_" + cfg, "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                lines.append("%-72s %7d %12.4f %12.2f %7.2f" % (short(r["Name"])[:72], int(r["Calls"]), float(r["AverageNs"]) / 1e6,
                                                               float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
                durs[short(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e9)
        open(os.path.join(out, "r3_%s_kernel_stats_rocprofv3.txt" % cfg), "w").write("\n".join(lines) + "\n")
        per = defaultdict(lambda: defaultdict(float))
        ndisp = defaultdict(set)
        for f in glob.glob(os.path.join(out, "pmc_mfma_" + cfg, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                per[k][row["Counter_Name"]] += float(row["Counter_Value"])
                ndisp[k].add(row["Dispatch_Id"])
        txt = ["# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES "
               "SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python tools/bench_configs.py %s  (%s)" % (cfg, label),
               "# kernels that issue FP64 MFMA instructions; flop = MOPS_F64 x 512 (the counter's unit) and, as a cross-check, "
               "instructions x 2048 (v_mfma_f64_16x16x4f64: 16 x 16 x 4 x 2 per wave instruction);",
               "# time = total duration of the same kernel in the --stats run above (same command); peak FP64 matrix = 78.6 TFLOP/s",
               "%-60s %7s %14s %14s %12s %10s %9s %9s" % ("kernel", "disp", "mfma_f64_inst", "MOPS_F64", "busy_cycles",
                                                         "time_ms", "TFLOP/s", "of peak")]
        for k, c in sorted(per.items()):
            inst, mops = c.get("SQ_INSTS_VALU_MFMA_F64", 0.0), c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
            if inst == 0 and mops == 0:
                continue
            calls, tsec = durs.get(k, (0, 0.0))
            flop = mops * 512.0 if mops else inst * 2048.0
            tf = flop / tsec / 1e12 if tsec else float("nan")
            txt.append("%-60s %7d %14.0f %14.0f %12.0f %10.3f %9.2f %9.3f" % (k[:60], len(ndisp[k]), inst, mops,
                                                                          c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
                                                                          1e3 * tsec, tf, tf / 78.6))
        open(os.path.join(out, "r3_%s_mfma_counters.txt" % cfg), "w").write("\n".join(txt) + "\n")
        print("\n".join(lines[:14]))
        print("\n".join(txt))


if __name__ == "__main__":
    main()
