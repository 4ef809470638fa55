#!/bin/bash
# full GPU suite + bench (driver command line) + smoke
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3v_pytest.txt 2>&1
tail -5 gpurun_out/r3v_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3v_smoke.txt 2>&1; tail -2 gpurun_out/r3v_smoke.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r3_full.json 2> gpurun_out/bench_r3_full.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_r3_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["checksum_b_c_l2"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
PY
