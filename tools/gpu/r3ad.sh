#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pencil.py tests/test_gpu_ivp.py -x -q -m gpu -k "window_matvec or zero_row" 2>&1 | tail -5
