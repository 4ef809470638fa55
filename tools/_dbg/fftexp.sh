mkdir -p gpurun_out
run() { echo "=== $*"; env "$@" python tools/bench_kernels.py 2>&1 | grep -v "torch copy"; }
run A=0
run DDH_FFT_PROF=1
run DDH_FFT_TWDIRECT=1
run DDH_FFT_RADIX=8,8,4,3
run DDH_FFT_RADIX=4,4,4,4,3
run DDH_FFT_RADIX=16,8,3
run DDH_FFT_RADIX=8,4,4,3
run DDH_FFT_RADIX=3,16,16
run DDH_FFT_RADIX=3,8,16
