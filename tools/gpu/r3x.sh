#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_wave_transforms.py -x -q -m gpu 2>&1 | tail -2
python tools/bench_strided.py 2>&1 | grep -v "^#" | tail -8
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bx.json 2> gpurun_out/bx.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bx.json").read().strip().splitlines()[-1])
k=d["kernels"]
print(d["value"], d["ms_per_step"], d["checksum_b_c_l2"], {n: round(k[n]["avg_ms"],3) for n in k if "rfft" in n}, d["parity"]["max_residual"])
PY
