"""Register budget of the hot kernels, read from the compiler (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).

The strided wave transforms and the fused grid stage run at 2 waves per SIMD with up to 256 registers: one value too many
and the compiler spills into scratch memory INSIDE the transform loop -- the dual x transform did (60 bytes per lane), which
showed up as 1.16 x its algorithmic HBM traffic in the PMC counters and 7 % of its time.  This test
keeps every kernel of the headline step free of scratch (the backward sweep's 32 bytes are loop-invariant values saved once
per thread: allowed, bounded)."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dedalus_amd", "csrc")


def _usage(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "--cuda-device-only", "-c",
           os.path.join(CSRC, src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        for key, tag in (("VGPRs", "vgprs"), (r"ScratchSize \[bytes/lane\]", "scratch"), (r"Occupancy \[waves/SIMD\]", "waves")):
            m = re.search(key + r": (\d+)", line)
            if m and name:
                out[name][tag] = int(m.group(1))
    assert out, r.stderr[-2000:]
    return out


def test_strided_wave_transforms_do_not_spill():
    u = _usage("ddh_fftwave.hip")
    hot = {k: v for k, v in u.items() if "wave_rfft_kernel" in k or "wave_cheb_kernel" in k}
    assert len(hot) >= 10
    for k, v in hot.items():
        assert v["scratch"] == 0, (k, v)
        assert v["waves"] >= 2, (k, v)


def test_fused_grid_stage_does_not_spill():
    u = _usage("ddh_gridwave2.hip")
    # the default instance: C = 6 (N = 768), 4 operand blocks, 4 waves, twiddles in registers, operands by LDS-DMA
    k = [n for n in u if "gridwave2_bilinear_kernelILi6ELi4ELi4ELb1ELb1E" in n]
    assert len(k) == 1, list(u)
    assert u[k[0]]["scratch"] == 0 and u[k[0]]["vgprs"] <= 256 and u[k[0]]["waves"] >= 2, u[k[0]]
