mkdir -p gpurun_out/r3c
for tpw in 1 2 4 8; do for ws in 0 1; do
  DDH_FFT_TPW=$tpw DDH_FFT_WSYNC=$ws python tools/bench_strided.py > gpurun_out/r3c/bench_tpw${tpw}_ws${ws}.txt 2>&1
done; done
tail -n 8 gpurun_out/r3c/bench_*.txt
for v in 0 1 3; do
DDH_FFT_WAVE=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c/full_wave$v.json 2> gpurun_out/r3c/full_wave$v.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3c/full_wave$v.json").read().strip().splitlines()[-1])
print("WAVE=$v", d["value"], d["ms_per_step"], d["checksum_b_c_l2"], {k:(round(x["avg_ms"],3)) for k,x in d["kernels"].items()})
PY
done
