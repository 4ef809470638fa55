#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bo_$tag.json 2> gpurun_out/bo_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bo_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d["checksum_b_c_l2"], d["kernels"]["pencil_solve"]["avg_ms"], d["kernels"]["cheb_backward_strided_dual"]["avg_ms"])
except Exception as e:
    print("$tag failed", e)
PY
}
run base A=1
run nts DDH_LIB=$GRAFT_REPO_ROOT/dedalus_amd/csrc/libvariant_nts.so
run ntsl DDH_LIB=$GRAFT_REPO_ROOT/dedalus_amd/csrc/libvariant_ntsl.so
