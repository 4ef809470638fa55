#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bp_$tag.json 2> gpurun_out/bp_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bp_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d["checksum_b_c_l2"], d["kernels"]["rfft_bilinear_fused"]["avg_ms"])
except Exception as e:
    print("$tag failed", e)
PY
}
L=$GRAFT_REPO_ROOT/dedalus_amd/csrc
run base A=1
run e1 DDH_LIB=$L/libvariant_e1.so
run e2 DDH_LIB=$L/libvariant_e2.so
run e3 DDH_LIB=$L/libvariant_e3.so
run e3_lpw4 DDH_LIB=$L/libvariant_e3.so DDH_GW_LPW=4
run e3_lpw16 DDH_LIB=$L/libvariant_e3.so DDH_GW_LPW=16
run e123 DDH_LIB=$L/libvariant_e123.so
