#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "occ2        $(python tools/bench_fused.py 2>&1 | tail -1)"
echo "occ3        $(DDH_GW_OCC=3 python tools/bench_fused.py 2>&1 | tail -1)"
echo "occ2 notw   $(DDH_GW_TWREG=0 python tools/bench_fused.py 2>&1 | tail -1)"
echo "occ2 w8     $(DDH_GW_WAVES=8 python tools/bench_fused.py 2>&1 | tail -1)"
echo "occ3 lpw16  $(DDH_GW_OCC=3 DDH_GW_LPW=16 python tools/bench_fused.py 2>&1 | tail -1)"
