mkdir -p gpurun_out/r3h
for sz in 256,512,256 128,512,256 64,512,256; do
  python bench.py --size $sz --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/r3h/share_$sz.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3h/share_$sz.json").read().strip().splitlines()[-1])
print("$sz", round(d["ms_per_step"],2), {k:(round(x["avg_ms"],3), x["launches"]) for k,x in d["kernels"].items()})
PY
done
# solve variants at the P = 8 and P = 4 shares
for sz in 64,512,256 128,512,256; do for v in "DDH_COOP_FWD=1 DDH_COOP_CB=16" "DDH_COOP_FWD=1 DDH_COOP_CB=4" "DDH_COOP_FWD=0 DDH_COOP_CB=4" "DDH_COOP_FWD=0 DDH_COOP_CB=0"; do
  env $v python bench.py --size $sz --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/r3h/tmp.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3h/tmp.json").read().strip().splitlines()[-1])
print("$sz", "$v", round(d["ms_per_step"],2), "solve", round(d["kernels"]["pencil_solve"]["avg_ms"],3))
PY
done; done
python -m pytest tests/test_gpu_comm.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
