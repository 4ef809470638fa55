#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bn_$tag.json 2> gpurun_out/bn_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bn_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d.get("state_checksum"), d["kernels"]["pencil_solve"], d["parity"]["max_residual"], d["parity"]["max_solution_error"])
except Exception as e:
    print("$tag failed", e)
PY
}
run tile4 DDH_PAIR_TILE8=0
run tile8 DDH_PAIR_TILE8=1
python -m pytest tests/test_gpu_pencil.py tests/test_gpu_baseline_sizes.py -x -q -m gpu 2>&1 | tail -3
