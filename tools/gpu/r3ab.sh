#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bab_$tag.json 2> gpurun_out/bab_$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bab_$tag.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$tag", d["value"], d["ms_per_step"], repr(d["checksum_b_c_l2"]), k["pencil_solve"], d["parity"]["max_residual"], d["parity"]["max_solution_error"])
PY
}
run zrows A=1
run nozrows DDH_NO_ZERO_ROWS=1
python -m pytest tests/test_gpu_pencil.py tests/test_gpu_ivp.py tests/test_gpu_reference_pencils.py tests/test_gpu_baseline_sizes.py -x -q -m gpu 2>&1 | tail -2
