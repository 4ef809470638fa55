#!/bin/bash
# timing ablations of the backward sweep (DDH_BWD_DEEP = 100 + mask: 1 one factor load per row, 2 no FMAs, 4 no stores,
# 8 no scratch load, 16 no window shift, 32 stores to lane-contiguous addresses); the solve family = forward 4.5 ms + backward
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/bk_$tag.json 2> gpurun_out/bk_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bk_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], d["kernels"]["pencil_solve"], d["kernels"]["pencil_matvec"])
except Exception as e:
    print("$tag failed", e)
PY
}
run base DDH_BWD_DEEP=0
run contig DDH_BWD_DEEP=132
run nopfuse DDH_NO_PFUSE=1
run nopair DDH_PAIR=0
run nopair_contig DDH_PAIR=0 DDH_BWD_DEEP=132
run nopair_nostore DDH_PAIR=0 DDH_BWD_DEEP=104
