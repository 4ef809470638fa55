#!/bin/bash
cd $GRAFT_REPO_ROOT
for sz in 64,512,256 128,512,256; do
  python bench.py --size $sz --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/by_$sz.json 2> gpurun_out/by_$sz.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/by_$sz.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$sz", d["ms_per_step"], "solve", k["pencil_solve"]["avg_ms"], d["parity"]["max_residual"], d["parity"]["max_solution_error"])
PY
done
python -m pytest tests/test_gpu_pencil.py -x -q -m gpu 2>&1 | tail -2
