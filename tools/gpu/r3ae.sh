#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bae_$tag.json 2> gpurun_out/bae_$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bae_$tag.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$tag", d["value"], d["ms_per_step"], repr(d["checksum_b_c_l2"]), k["pencil_solve"]["avg_ms"], d["parity"]["max_residual"], d["parity"]["max_solution_error"])
PY
}
run skip A=1
run noskip DDH_NO_SKIP_ROWS=1
python -m pytest tests/test_gpu_ivp.py tests/test_gpu_baseline_sizes.py tests/test_gpu_output.py -x -q -m gpu 2>&1 | tail -2
