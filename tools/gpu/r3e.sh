mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_dense_inverse.py tests/test_gpu_boundary.py tests/test_gpu_pencil.py tests/test_gpu_ivp.py tests/test_gpu_sphere.py -x -q -m gpu > gpurun_out/r3e/tests.txt 2>&1
grep -E "passed|failed|error|Error|assert" gpurun_out/r3e/tests.txt | tail -15
python tools/time_refactor.py > gpurun_out/r3e/refactor.txt 2>&1; tail -3 gpurun_out/r3e/refactor.txt
DDH_FLAG_HOST_INV=1 python tools/time_refactor.py > gpurun_out/r3e/refactor_hostinv.txt 2>&1; tail -3 gpurun_out/r3e/refactor_hostinv.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cfl > gpurun_out/r3e/bench_cfl.json 2> gpurun_out/r3e/bench_cfl.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3e/bench_cfl.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["checksum_b_c_l2"]); print(d.get("cfl_mode")); print(d["parity"])
PY
tail -5 gpurun_out/r3e/bench_cfl.err
