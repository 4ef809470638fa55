mkdir -p gpurun_out/r3j
python -m pytest tests -x -q -m gpu > gpurun_out/r3j/tests.txt 2>&1
grep -E "passed|failed|rror|assert" gpurun_out/r3j/tests.txt | tail -12
