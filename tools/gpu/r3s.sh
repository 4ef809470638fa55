#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in 32 4 8 16 64 128; do
  DDH_MV_CHUNKS=$c python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/bs_$c.json 2> gpurun_out/bs_$c.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bs_$c.json").read().strip().splitlines()[-1])
print("chunks=$c", d["ms_per_step"], d["kernels"]["pencil_matvec"])
PY
done
