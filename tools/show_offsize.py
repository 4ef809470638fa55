"""Table of tools/bench_configs.py --offsize / --json output: steps/s and every kernel family's fraction of the HBM peak."""
import json
import sys

for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("%-44s %9.2f steps/s  %8.3f ms/step  wave transforms/step %.0f%s" % (
        d["config"], d["steps_per_s"], d["ms_per_step"], d.get("wave_kernel_transforms_per_step", 0),
        ("  binv residual %.1e" % d["block_inverse_residual_max"]) if "block_inverse_residual_max" in d else ""))
    for k, v in sorted(d.get("kernel_families", {}).items(), key=lambda kv: -kv[1]["ms_per_launch"] * kv[1]["launches_per_step"]):
        print("      %-30s %5.1f /step  %8.4f ms  %7.2f GB  %7.0f GB/s  frac %.2f" % (
            k, v["launches_per_step"], v["ms_per_launch"], v["algorithmic_GB_per_launch"], v["GBps"], v["frac_of_hbm_peak"]))
