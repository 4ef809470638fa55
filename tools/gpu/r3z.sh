#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids
python tools/bench_fused.py 2>&1 | tail -1
