#!/bin/bash
# A/B of the deep-prefetch backward sweep
cd $GRAFT_REPO_ROOT
for d in 0 3 4; do
  DDH_BWD_DEEP=$d python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bk_deep$d.json 2> gpurun_out/bk_deep$d.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bk_deep$d.json").read().strip().splitlines()[-1])
print("deep=$d", d["value"], d["ms_per_step"], d.get("checksum"), d["kernels"]["pencil_solve"], d.get("parity"))
PY
done
