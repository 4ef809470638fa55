# Clock / power of the GPU during the timed steps of bench.py (rocm-smi sampled beside it; the step is ~46 ms, a sample is an
# average of the management firmware over its own window)
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/r5_step_clocks.txt
: > $OUT
python bench.py --steps 400 --warmup 5 --no-cpu-baseline > gpurun_out/r5_step_clocks_bench.json 2> gpurun_out/r5_step_clocks_bench.err &
PID=$!
sleep 22
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|Graphics Package Power" | tr '\n' ' ' | sed 's/GPU\[0\]\t\t: //g' >> $OUT
  echo >> $OUT
  sleep 0.7
done
wait $PID
python - <<'PY' >> $OUT
import json
d=json.loads(open("gpurun_out/r5_step_clocks_bench.json").read().strip().splitlines()[-1])
print("bench: steps/s %.3f ms %.2f"%(d["value"],d["ms_per_step"]))
PY
cat $OUT
