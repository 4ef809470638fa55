#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/baf.json 2> gpurun_out/baf.err
python - <<PY
import json
d=json.loads(open("gpurun_out/baf.json").read().strip().splitlines()[-1])
k=d["kernels"]
print(d["value"], d["ms_per_step"], repr(d["checksum_b_c_l2"]), k["pencil_matvec"]["avg_ms"], k["pencil_solve"]["avg_ms"], d["parity"]["max_residual"], d["parity"]["max_solution_error"])
PY
python -m pytest tests/test_gpu_ivp.py tests/test_gpu_pencil.py tests/test_gpu_baseline_sizes.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -2
