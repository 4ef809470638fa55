#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/dedalus_amd/csrc
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/bt_$tag.json 2> gpurun_out/bt_$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bt_$tag.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$tag", d["ms_per_step"], d["checksum_b_c_l2"], {n: round(k[n]["avg_ms"],3) for n in k if "rfft" in n})
PY
}
run w8 A=1
run w4 DDH_LIB=$L/libvariant_w4.so
