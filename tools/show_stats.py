"""Print the per-kernel table of a rocprofv3 --kernel-trace --stats output directory."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
        print("%-72s %6s %9.4f ms avg %9.2f ms total %6.2f%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                             float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
