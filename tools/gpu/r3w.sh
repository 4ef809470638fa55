#!/bin/bash
cd $GRAFT_REPO_ROOT
for sz in 512,512,256 512,480,256 480,512,256 480,480,256; do
  python bench.py --size $sz --steps 3 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/bw_$sz.json 2> gpurun_out/bw_$sz.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bw_$sz.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$sz", d["ms_per_step"], "matvec", k["pencil_matvec"]["avg_ms"], k["pencil_matvec"]["GBps"], "solve", k["pencil_solve"]["avg_ms"], k["pencil_solve"]["GBps"])
PY
done
