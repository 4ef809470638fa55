# A/B of two builds of the library on the same box: dedalus_amd/csrc/libdedalus_hip_base.so (DDH_LIB) against the in-tree build
mkdir -p gpurun_out/r5ab
python -m pytest tests/test_gpu_wave_transforms.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2; do
DDH_LIB=$PWD/dedalus_amd/csrc/libdedalus_hip_base.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5ab/base$i.json 2> gpurun_out/r5ab/base$i.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5ab/new$i.json 2> gpurun_out/r5ab/new$i.err
done
python - <<'PY'
import json
for n in ("base1","new1","base2","new2"):
    try:
        d=json.loads(open("gpurun_out/r5ab/%s.json"%n).read().strip().splitlines()[-1])
        print(n, "steps/s %.3f ms %.2f chk %.15g frac %.3f"%(d["value"],d["ms_per_step"],d["checksum_b_c_l2"],d["roofline"]["frac"]))
        print("   ", {k:(round(v["avg_ms"],3)) for k,v in d["kernels"].items()})
    except Exception as e:
        print(n,"unreadable",e); print(open("gpurun_out/r5ab/%s.err"%n).read()[-2000:])
PY
