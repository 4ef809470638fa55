#!/bin/bash
# fused y stage: LDS ablations (DDH_FFT_DBG bits: 1 no butterfly math, 4 no global loads, 16 no global stores,
# 64 no LDS writes, 128 no LDS reads; 2 = the ablation instance with nothing disabled)
cd $GRAFT_REPO_ROOT
for d in 0 2 66 130 194 195 215; do
  echo "dbg=$d $(DDH_FFT_DBG=$d python tools/bench_fused.py 2>&1 | tail -1)"
done
