"""Race screen of the fused y stage (LDS-DMA staging, hand-issued loads, two operands in flight): the SAME launch repeated many
times on the same operands must give bit-identical results every time, and equal the first-generation kernel's to round-off
(run once with DDH_GW_V2=0 to write the reference, then with the default to compare).
    DDH_GW_V2=0 python tools/gpu/fused_stress.py write /tmp/ref.pt;  python tools/gpu/fused_stress.py check /tmp/ref.pt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dedalus_amd.device import Device  # noqa: E402
from dedalus_amd.executor import HipExecutor  # noqa: E402


def main():
    mode, path = sys.argv[1], sys.argv[2]
    reps = int(os.environ.get("STRESS_REPS", "60"))
    dev = Device.get()
    t = dev.torch
    t.manual_seed(1234)
    Ny, Nz = 512, 256
    Gx, Gy, Gz = 3 * Ny // 2, 3 * Ny // 2, 3 * Nz // 2
    hx = HipExecutor(dev)
    nl = (Gz // int(os.environ.get("FUSED_ZDIV", "2"))) * Gx
    a = t.randn((3, nl, Ny), dtype=t.float64, device=dev.tdev)
    bb = t.randn((12, nl, Ny), dtype=t.float64, device=dev.tdev)
    a[..., 1] = 0.0
    bb[..., 1] = 0.0
    terms = [(0, j, j, 1.0) for j in range(3)] + [(1 + c, j, 3 + 3 * j + c, 1.0) for c in range(3) for j in range(3)]
    bds = [0.0] * 12
    for i in (1, 6, 7, 8):
        bds[i] = 1.5707963267948966
    first = None
    bad = 0
    for r in range(reps):
        oo = dev.empty((4, nl, Ny))
        oo.fill_(float("nan"))
        hx.rfft_bilinear_fused(("rfft", Gy, Ny), None, [a[i] for i in range(3)], [bb[i] for i in range(12)],
                               [oo[i] for i in range(4)], nl, terms, b_dscale=bds)
        dev.sync()
        if first is None:
            first = oo.clone()
        elif not t.equal(first, oo):
            bad += 1
    print("%d launches, %d differ from the first" % (reps, bad))
    assert bad == 0
    if mode == "write":
        t.save(first.cpu(), path)
    else:
        ref = t.load(path).to(first.device)
        if ref.shape != first.shape:                 # (another FUSED_ZDIV than the reference run: repeatability only)
            return
        err = float((first - ref).abs().max() / ref.abs().max())
        print("max |difference| to the reference kernel / max |reference| = %.3e" % err)
        assert err < 1e-13


if __name__ == "__main__":
    main()
