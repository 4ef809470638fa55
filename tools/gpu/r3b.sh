set -x
mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_wave_transforms.py tests/test_gpu_transforms.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3b/tests.txt
cat gpurun_out/r3b/tests.txt
python tools/bench_strided.py > gpurun_out/r3b/bench_default.txt 2>&1
NCOMP=1 python tools/bench_strided.py > gpurun_out/r3b/bench_nc1.txt 2>&1
DDH_FFT_TPW=2 python tools/bench_strided.py > gpurun_out/r3b/bench_tpw2.txt 2>&1
DDH_FFT_TPW=4 python tools/bench_strided.py > gpurun_out/r3b/bench_tpw4.txt 2>&1
DDH_FFT_TPW=16 python tools/bench_strided.py > gpurun_out/r3b/bench_tpw16.txt 2>&1
cat gpurun_out/r3b/bench_*.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3b/bench_full.json 2> gpurun_out/r3b/bench_full.err
tail -c 3000 gpurun_out/r3b/bench_full.json
