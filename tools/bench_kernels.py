"""Micro-benchmark of the transform kernels at the 3-D Rayleigh-Benard line sizes (one component).
Prints achieved algorithmic GB/s = (bytes read + bytes written) / time, measured with HIP events
on the launch stream."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedalus_amd import libhip  # noqa: E402
from dedalus_amd.device import Device, ptr  # noqa: E402


def plan(name, *args):
    h = C.c_uint64(0)
    libhip.call(name, C.byref(h), *args)
    return h


def timeit(dev, fn, reps=5):
    t = dev.torch
    fn()
    dev.sync()
    e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    dev.sync()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = Device.get()
    t = dev.torch
    Nx = Ny = int(os.environ.get("NXY", "512"))
    Nz = int(os.environ.get("NZ", "256"))
    Gx, Gy, Gz = 3 * Nx // 2, 3 * Ny // 2, 3 * Nz // 2
    cases = [
        # name, kind, N, M, coeff shape [outer, M, inner], as seen by the kernel
        ("z  cheb  bwd/fwd  [kz,kx*ky]", "cheb", Gz, Nz, (1, Nz, Nx * Ny)),
        ("y  rfft  bwd/fwd  contiguous", "rfft", Gy, Ny, (Gz * Nx, Ny, 1)),
        ("x  rfft  bwd/fwd  inner=Ny  ", "rfft", Gx, Nx, (Gz, Nx, Ny)),
    ]
    for name, kind, N, M, cs in cases:
        p = plan("ddh_plan_cheb", N, M, 0, None, None) if kind == "cheb" else plan("ddh_plan_rfft", N, M)
        c = t.randn(cs, dtype=t.float64, device=dev.tdev)
        gs = (cs[0], N, cs[2])
        g = dev.empty(gs)
        c2 = dev.empty(cs)
        bytes_ = (c.numel() + g.numel()) * 8
        tb = timeit(dev, lambda: libhip.call("ddh_%s_backward" % kind, p, ptr(c), ptr(g), cs[0], cs[2], dev.stream))
        tf = timeit(dev, lambda: libhip.call("ddh_%s_forward" % kind, p, ptr(g), ptr(c2), cs[0], cs[2], dev.stream))
        if os.environ.get("DDH_FFT_PROF"):
            lib = libhip.load()
            lib.ddh_debug_fft_prof.argtypes = [C.c_uint64, C.POINTER(C.c_double)]
            out = (C.c_double * 4)()
            # counters accumulated over both directions' launches: reset, then one bwd and one fwd
            lib.ddh_debug_fft_prof(p, out)
            libhip.call("ddh_%s_backward" % kind, p, ptr(c), ptr(g), cs[0], cs[2], dev.stream); dev.sync()
            lib.ddh_debug_fft_prof(p, out); b4 = list(out)
            libhip.call("ddh_%s_forward" % kind, p, ptr(g), ptr(c2), cs[0], cs[2], dev.stream); dev.sync()
            lib.ddh_debug_fft_prof(p, out); f4 = list(out)
            print("   phase cycles/WG  bwd: load %.0f fft %.0f store %.0f | fwd: load %.0f fft %.0f store %.0f  (WGs %d)"
                  % (b4[0], b4[1], b4[2], f4[0], f4[1], f4[2], int(b4[3])))
        print("%s  bwd %.3f ms %.0f GB/s | fwd %.3f ms %.0f GB/s   (%.2f GB/pass)" %
              (name, tb * 1e3, bytes_ / tb / 1e9, tf * 1e3, bytes_ / tf / 1e9, bytes_ / 1e9), flush=True)
        del c, g, c2
    # fused grid stage along y: u.grad(b) and u.grad(u) (3 + 12 operands, 4 results)
    from dedalus_amd.executor import HipExecutor
    hx = HipExecutor(dev)
    nl = (Gz // int(os.environ.get("FUSED_ZDIV", "4"))) * Gx
    padl = int(os.environ.get("FUSED_PAD", "0"))      # extra lines between components (HBM channel spread test)
    a = t.randn((3, nl + padl, Ny), dtype=t.float64, device=dev.tdev)[:, :nl]
    bb = t.randn((12, nl + padl, Ny), dtype=t.float64, device=dev.tdev)[:, :nl]
    oo = dev.empty((4, nl + padl, Ny))[:, :nl]
    terms = [(0, j, j, 1.0) for j in range(3)] + [(1 + c, j, 3 + 3 * j + c, 1.0) for c in range(3) for j in range(3)]
    tfz = timeit(dev, lambda: hx.rfft_bilinear_fused(("rfft", Gy, Ny), None, [a[i] for i in range(3)],
                                                     [bb[i] for i in range(12)], [oo[i] for i in range(4)], nl, terms))
    if os.environ.get("DDH_FFT_PROF"):
        lib = libhip.load()
        lib.ddh_debug_fft_prof.argtypes = [C.c_uint64, C.POINTER(C.c_double)]
        out = (C.c_double * 4)()
        ph = hx._plan(("rfft", Gy, Ny), None)[1]
        lib.ddh_debug_fft_prof(ph, out)
        hx.rfft_bilinear_fused(("rfft", Gy, Ny), None, [a[i] for i in range(3)], [bb[i] for i in range(12)],
                               [oo[i] for i in range(4)], nl, terms)
        dev.sync()
        lib.ddh_debug_fft_prof(ph, out)
        vals = [out[0] * out[3], out[1] * out[3], out[2] * out[3], out[3]]    # the hook divides by slot 3
        tot = sum(vals)
        nwg = (nl + 1) // 2
        print("   fused phase clocks per WG: wait %.0f unpack %.0f fft %.0f rest %.0f  -> %% %s"
              % tuple([v / nwg for v in vals] + [[round(100 * x / tot, 1) for x in vals]]))
    nbytes = 19 * nl * Ny * 8
    print("y  fused grid stage (19 lines arrays, %d lines): %.3f ms %.0f GB/s -> full size %.2f ms"
          % (nl, tfz * 1e3, nbytes / tfz / 1e9, tfz * 1e3 * Gz * Gx / nl), flush=True)
    del a, bb, oo
    # spin-weighted spherical harmonic colatitude transform at the sphere config (512 x 256 -> Lmax 254,
    # Ntheta 384): 255 matrices, 2 columns (cos / msin parts) -> batched GEMV bound by the matrix stream
    if os.environ.get("BENCH_SWSH", "1") != "0":
        from dedalus_amd.core.curvilinear import SWSHColatitudeTransform
        Lmax, Nth = 254, 384
        rows = [(m, 2 * m, 2 * m, 2, 0, 1, Lmax + 1 - m) for m in range(Lmax + 1)]
        t0 = __import__("time").time()
        sw = SWSHColatitudeTransform(Nth, Lmax, np.array(rows, dtype=np.int64), 0, executor=hx)
        tbuild = __import__("time").time() - t0
        for n3 in (1, 192):
            gg = t.randn((1, 2 * (Lmax + 1), Nth, n3), dtype=t.float64, device=dev.tdev)
            cc = dev.zeros((1, 2 * (Lmax + 1), Lmax + 1, n3))
            tfw = timeit(dev, lambda: sw.forward_reduced(gg, cc), reps=20)
            tbw = timeit(dev, lambda: sw.backward_reduced(cc, gg), reps=20)
            mat_bytes = sum((Lmax + 1 - m) * Nth * 8 for m in range(Lmax + 1))
            flops = 2.0 * sum((Lmax + 1 - m) * Nth for m in range(Lmax + 1)) * 2 * n3
            print("SWSH colatitude Lmax=254 Ntheta=384, %3d columns/m: fwd %.3f ms, bwd %.3f ms  (matrices %.0f MB -> "
                  "%.0f GB/s fwd; %.2f TFLOP/s fwd; host matrix build %.1f s)"
                  % (2 * n3, tfw * 1e3, tbw * 1e3, mat_bytes / 1e6, mat_bytes / tfw / 1e9, flops / tfw / 1e12, tbuild),
                  flush=True)
    # plain copy for reference
    a = t.randn(2 ** 27, dtype=t.float64, device=dev.tdev)
    b = t.empty_like(a)
    tc = timeit(dev, lambda: b.copy_(a))
    print("torch copy 1 GiB: %.0f GB/s" % (2 * a.numel() * 8 / tc / 1e9))


if __name__ == "__main__":
    main()
