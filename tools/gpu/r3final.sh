#!/bin/bash
# end of round 3: full GPU suite, smoke, bench at the driver's command line, the adaptive-dt mode, profiles
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3final_pytest.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r3final_pytest.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r3_final.json 2> gpurun_out/bench_r3_final.err
python bench.py --steps 10 --warmup 3 --cfl --no-cpu-baseline > gpurun_out/bench_r3_cfl.json 2> gpurun_out/bench_r3_cfl.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_r3_final.json").read().strip().splitlines()[-1])
print("final", d["value"], d["ms_per_step"], d["checksum_b_c_l2"], d["roofline"]["frac"], d["roofline"]["achieved"], d["roofline"]["traffic"])
c=json.loads(open("gpurun_out/bench_r3_cfl.json").read().strip().splitlines()[-1])
print("cfl", c.get("cfl_mode"))
PY
bash tools/profile_round3.sh > gpurun_out/prof_r3_run.log 2>&1
tail -2 gpurun_out/prof_r3_run.log
