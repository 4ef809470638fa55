# Clock / power of the GPU while the fused grid stage runs on random, smooth and all-zero operands (rocm-smi sampled beside it)
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/r5_fused_clocks.txt
: > $OUT
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4 >> $OUT
for data in randn smooth zeros; do
  echo "== FUSED_DATA=$data" >> $OUT
  FUSED_DATA=$data FUSED_REPS=4000 python tools/bench_fused.py >> $OUT 2>&1 &
  PID=$!
  sleep 12
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Power\|Graphics Package Power" | tr '\n' ' ' >> $OUT
    echo >> $OUT
    sleep 1
  done
  wait $PID
done
cat $OUT
