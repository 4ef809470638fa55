mkdir -p gpurun_out/r3d
python tools/bench_strided.py > gpurun_out/r3d/bench_default.txt 2>&1
cat gpurun_out/r3d/bench_default.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r3d/bench_full.json 2> gpurun_out/r3d/bench_full.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3d/bench_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["checksum_b_c_l2"], {k:(round(x["avg_ms"],3)) for k,x in d["kernels"].items()})
print(d["roofline"]); print(d["parity"])
PY
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
