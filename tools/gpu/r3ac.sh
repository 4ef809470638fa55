#!/bin/bash
# does the wave-per-four-pairs z kernel depend on the plane stride?  (same data volume, rows 2 MiB ... 4 KiB apart)
cd $GRAFT_REPO_ROOT
for inner in 262144 65536 16384 4096 512; do
  echo "inner=$inner (row stride $((inner*8/1024)) KiB): $(PYTHONPATH=. python tools/time_cheb_dual.py 3 $inner 2>&1 | grep -v amdgpu | tr '\n' ' ')"
done
