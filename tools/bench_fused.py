"""Timing of the fused grid stage alone at the 3-D Rayleigh-Benard line sizes (u.grad(b), u.grad(u):
3 + 12 operands, 4 results), for kernel experiments.  Environment knobs are read by the library:
DDH_FFT_DBG (ablations), DDH_GW_WAVES (waves per workgroup), DDH_FUSED_OLD=1 (workgroup-per-pair kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedalus_amd.device import Device  # noqa: E402
from dedalus_amd.executor import HipExecutor  # noqa: E402


def main():
    dev = Device.get()
    t = dev.torch
    Ny = int(os.environ.get("NXY", "512"))
    Nz = int(os.environ.get("NZ", "256"))
    Gx, Gy, Gz = 3 * Ny // 2, 3 * Ny // 2, 3 * Nz // 2
    hx = HipExecutor(dev)
    nl = (Gz // int(os.environ.get("FUSED_ZDIV", "1"))) * Gx
    a = t.randn((3, nl, Ny), dtype=t.float64, device=dev.tdev)
    bb = t.randn((12, nl, Ny), dtype=t.float64, device=dev.tdev)
    oo = dev.empty((4, nl, Ny))
    # FUSED_DATA=zeros / smooth: the time of the kernel follows its data through the power management (see
    # profiles/r5_fused_variants.txt): all-zero operands, or spectra that decay like the fields of a resolved flow
    data = os.environ.get("FUSED_DATA", "randn")
    if data == "zeros":
        a.zero_()
        bb.zero_()
    elif data == "smooth":
        decay = t.exp(-t.arange(Ny, dtype=t.float64, device=dev.tdev) / (Ny / 16.0))
        a *= decay
        bb *= decay
    terms = [(0, j, j, 1.0) for j in range(3)] + [(1 + c, j, 3 + 3 * j + c, 1.0) for c in range(3) for j in range(3)]

    # FUSED_DERIV=1: four of the operands differentiated at load, as in the step (d/dy of u and b)
    bds = None
    if os.environ.get("FUSED_DERIV", "0") == "1":
        bds = [0.0] * 12
        for i in (1, 6, 7, 8):
            bds[i] = 1.5707963267948966

    def run():
        hx.rfft_bilinear_fused(("rfft", Gy, Ny), None, [a[i] for i in range(3)], [bb[i] for i in range(12)],
                               [oo[i] for i in range(4)], nl, terms, b_dscale=bds)
    run()
    dev.sync()
    e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("FUSED_REPS", "5"))
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    dev.sync()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 19 * nl * Ny * 8
    chk = [float(oo[i].abs().sum().item()) for i in range(4)]
    print("data=%s " % data, end="")
    print("fused grid stage %d lines  dbg=%s waves=%s old=%s v1=%s twreg=%s lpw=%s: %.3f ms  %.0f GB/s  checksums %s" % (
        nl, os.environ.get("DDH_FFT_DBG", "0"), os.environ.get("DDH_GW_WAVES", "4"),
        os.environ.get("DDH_FUSED_OLD", "0"), os.environ.get("DDH_GW_V1", "0"), os.environ.get("DDH_GW_TWREG", "1"),
        os.environ.get("DDH_GW_LPW", "-"), ms, nbytes / ms / 1e6, " ".join("%.15e" % c for c in chk)), flush=True)


if __name__ == "__main__":
    main()
