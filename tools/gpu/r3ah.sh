#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ivp.py tests/test_gpu_baseline_sizes.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bah.json 2> gpurun_out/bah.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bah.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], repr(d["checksum_b_c_l2"]), d["kernels"]["pencil_solve"]["avg_ms"], d["roofline"]["algorithmic_bytes_per_launch"], d["parity"]["max_residual"])
PY
