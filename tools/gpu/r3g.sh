mkdir -p gpurun_out/r3g
python -m pytest tests/test_gpu_pencil.py tests/test_gpu_reference_pencils.py tests/test_gpu_ivp.py -x -q -m gpu > gpurun_out/r3g/tests.txt 2>&1
grep -E "passed|failed|error|Error|assert" gpurun_out/r3g/tests.txt | tail -15
python tools/time_refactor.py > gpurun_out/r3g/refactor.txt 2>&1; tail -2 gpurun_out/r3g/refactor.txt
DDH_FACTOR_ROWS=0 python tools/time_refactor.py > gpurun_out/r3g/refactor_old.txt 2>&1; tail -2 gpurun_out/r3g/refactor_old.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cfl > gpurun_out/r3g/bench_cfl.json 2> gpurun_out/r3g/bench_cfl.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3g/bench_cfl.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["checksum_b_c_l2"]); print(d.get("cfl_mode")); print(d["parity"])
PY
