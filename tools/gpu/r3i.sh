mkdir -p gpurun_out/r3i
python -m pytest tests/test_gpu_transforms.py -x -q -m gpu -k fused 2>&1 | grep -E "passed|failed|error" | tail -3
for v in 1 0; do
DDH_GW_TWREG=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3i/bench_tw$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r3i/bench_tw$v.json").read().strip().splitlines()[-1])
print("TWREG=$v", round(d["value"],3), round(d["ms_per_step"],2), d["checksum_b_c_l2"], "fused", d["kernels"]["rfft_bilinear_fused"]["avg_ms"], "solve", d["kernels"]["pencil_solve"]["avg_ms"])
PY
done
