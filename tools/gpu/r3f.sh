mkdir -p gpurun_out/r3f
python -m pytest tests/test_gpu_shell.py -x -q -m gpu -s > gpurun_out/r3f/tests.txt 2>&1
grep -E "passed|failed|error|Error|assert|vs reference" gpurun_out/r3f/tests.txt | tail -15
python tools/time_refactor.py > gpurun_out/r3f/refactor.txt 2>&1; tail -3 gpurun_out/r3f/refactor.txt
DDH_FLAG_HOST_INV=1 python tools/time_refactor.py > gpurun_out/r3f/refactor_hostinv.txt 2>&1; tail -3 gpurun_out/r3f/refactor_hostinv.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r3f/prof_refactor -- python /root/repo/tools/time_refactor.py > /root/repo/gpurun_out/r3f/prof_refactor.log 2>&1
cd /root/repo; find gpurun_out/r3f/prof_refactor -name "*kernel_stats*" | head; f=$(find gpurun_out/r3f/prof_refactor -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
