#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_baseline_sizes.py -x -q -m gpu -k explicit -s > gpurun_out/r3r.txt 2>&1
grep -E "explicit half|passed|failed|Error|error" gpurun_out/r3r.txt | tail -12
python -m pytest tests/test_gpu_wave_transforms.py tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/r3r2.txt 2>&1
tail -3 gpurun_out/r3r2.txt
