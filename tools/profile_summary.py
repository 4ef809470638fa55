"""Condense the rocprofv3 outputs of tools/profile_round.sh into the files committed under profiles/:
  <tag>_kernel_stats_rocprofv3.txt   per-kernel calls / average / total duration (--kernel-trace --stats)
  <tag>_pmc_summary.txt, <tag>_pmc_traffic.json   per-kernel HBM bytes (FETCH_SIZE x2 on gfx950, WRITE_SIZE)
                                                  and SQ wait / LDS-conflict ratios
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    """Strip namespaces and argument lists from a demangled kernel name, keep template arguments."""
    name = re.sub(r"\(.*$", "", name).strip()
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("ddh::", "")
    return name


def read_counters(d):
    per = defaultdict(lambda: defaultdict(list))       # kernel -> counter -> [per-dispatch value]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        disp = defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            key = (row.get("Dispatch_Id"), row.get("Counter_Name"))
            disp[key] += float(row["Counter_Value"])
            names[row.get("Dispatch_Id")] = short(row["Kernel_Name"])
        for (did, cname), v in disp.items():
            per[names[did]][cname].append(v)
    return per


def main():
    out, tag = sys.argv[1], sys.argv[2]
    # ---- stats
    lines = []
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        lines.append("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "
                     "(MI355X, 3-D RB 512x512x256 RK222; build + 2 warm-up + 5 timed steps)")
        lines.append("%-64s %7s %12s %12s %7s" % ("kernel", "calls", "avg_ms", "total_ms", "%"))
        for r in rows:
            lines.append("%-64s %7d %12.4f %12.2f %7.2f" % (short(r["Name"])[:64], int(r["Calls"]),
                                                           float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6,
                                                           float(r["Percentage"])))
    open(os.path.join(out, "%s_kernel_stats_rocprofv3.txt" % tag), "w").write("\n".join(lines) + "\n")
    # ---- PMC
    sq = read_counters(os.path.join(out, "pmc_sq"))
    fe = read_counters(os.path.join(out, "pmc_fetch"))
    wr = read_counters(os.path.join(out, "pmc_write"))
    kernels = sorted(set(sq) | set(fe) | set(wr))
    traffic = {}
    txt = ["# rocprofv3 --kernel-trace --pmc <counters> (separate passes: SQ_*, FETCH_SIZE, WRITE_SIZE) -- "
           "python bench.py --steps 1 --warmup 1 --no-cpu-baseline",
           "# MI355X, 3-D RB 512x512x256 RK222.  FETCH_SIZE/WRITE_SIZE are in KiB; HBM read bytes = 2 x FETCH_SIZE x 1024 "
           "on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE x 1024 taken as is.",
           "# wait_any / wait_inst / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES; "
           "ldsconf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
           "%-58s %5s %11s %11s %11s %11s | %8s %8s %8s %8s" % ("kernel", "n", "rd_GB_mean", "rd_GB_max", "wr_GB_mean",
                                                              "wr_GB_max", "wait_any", "wait_ins", "active", "ldsconf")]
    for k in kernels:
        rd = [2.0 * 1024.0 * v / 1e9 for v in fe.get(k, {}).get("FETCH_SIZE", [])]
        ww = [1024.0 * v / 1e9 for v in wr.get(k, {}).get("WRITE_SIZE", [])]
        s = sq.get(k, {})
        wc = sum(s.get("SQ_WAVE_CYCLES", [])) or float("nan")
        la = sum(s.get("SQ_LDS_IDX_ACTIVE", []))
        n = max(len(rd), len(ww), len(s.get("SQ_WAVE_CYCLES", [])))
        mean = lambda a: sum(a) / len(a) if a else float("nan")
        txt.append("%-58s %5d %11.2f %11.2f %11.2f %11.2f | %8.2f %8.2f %8.2f %8.2f" % (
            k[:58], n, mean(rd), max(rd) if rd else float("nan"), mean(ww), max(ww) if ww else float("nan"),
            sum(s.get("SQ_WAIT_ANY", [])) / wc, sum(s.get("SQ_WAIT_INST_ANY", [])) / wc,
            sum(s.get("SQ_ACTIVE_INST_ANY", [])) / wc,
            (sum(s.get("SQ_LDS_BANK_CONFLICT", [])) / la) if la else 0.0))
        if rd or ww:
            traffic[k] = dict(read_GB_mean=mean(rd), write_GB_mean=mean(ww), launches=n)
    open(os.path.join(out, "%s_pmc_summary.txt" % tag), "w").write("\n".join(txt) + "\n")
    json.dump(traffic, open(os.path.join(out, "%s_pmc_traffic.json" % tag), "w"), indent=1)
    print("\n".join(lines[:40]))
    print("\n".join(txt))


if __name__ == "__main__":
    main()
