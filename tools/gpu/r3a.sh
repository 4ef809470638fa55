set -x
mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_wave_transforms.py tests/test_gpu_transforms.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3a/tests.txt
cat gpurun_out/r3a/tests.txt
python tools/bench_strided.py > gpurun_out/r3a/bench_default.txt 2>&1
DDH_FFT_WAVE=0 python tools/bench_strided.py > gpurun_out/r3a/bench_old.txt 2>&1
BENCH_X=0 DDH_FFT_TPW=4 python tools/bench_strided.py > gpurun_out/r3a/bench_tpw4.txt 2>&1
BENCH_X=0 DDH_FFT_TPW=16 python tools/bench_strided.py > gpurun_out/r3a/bench_tpw16.txt 2>&1
BENCH_X=0 DDH_FFT_TPW=32 python tools/bench_strided.py > gpurun_out/r3a/bench_tpw32.txt 2>&1
BENCH_X=0 NCOMP=1 python tools/bench_strided.py > gpurun_out/r3a/bench_nc1.txt 2>&1
cat gpurun_out/r3a/bench_*.txt
