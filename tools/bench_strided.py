"""Micro-benchmark of the strided-axis transforms at the 3-D Rayleigh-Benard shapes (512 x 512 x 256, dealias 3/2),
one launch as the solver issues it: z Chebyshev dual backward (field + d/dz), z forward into the (3/2, 3/2) basis,
x real-FFT backward (single, dual) and forward.  Prints ms and algorithmic GB/s (bytes read + written once).
Environment switches of the library (DDH_FFT_WAVE, DDH_FFT_TPW, ...) are read once per process: run one process
per variant.  NCOMP components per launch (default 3)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedalus_amd import libhip  # noqa: E402
from dedalus_amd.device import Device, ptr  # noqa: E402
from dedalus_amd.tools import jacobi  # noqa: E402


def plan(name, *args):
    h = C.c_uint64(0)
    libhip.call(name, C.byref(h), *args)
    return h


def cheb_plan(N, M, alpha):
    if alpha == 0:
        return plan("ddh_plan_cheb", N, M, 0, None, None)
    dense = jacobi.conversion_matrix(M, -0.5, -0.5, alpha - 0.5, alpha - 0.5).toarray()
    offs = np.array([o for o in range(M) if np.any(np.diagonal(dense, o) != 0)], dtype=np.int32)
    bands = np.zeros((len(offs), M))
    for d, o in enumerate(offs):
        bands[d, :M - o] = np.diagonal(dense, o)
    return plan("ddh_plan_cheb", N, M, len(offs), libhip.as_ip(offs), libhip.as_dp(bands))


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
    nc = int(os.environ.get("NCOMP", "3"))
    Gx, Gz = 3 * Nx // 2, 3 * Nz // 2
    tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("DDH_"))
    print("# variant:", tag or "default", flush=True)
    out = {}
    # ---- z axis: [comp][kz][kx*ky]
    inner = Nx * Ny
    c = t.randn((nc, Nz, inner), dtype=t.float64, device=dev.tdev)
    g = dev.empty((nc, Gz, inner))
    g2 = dev.empty((nc, Gz, inner))
    p0, p1, p2 = cheb_plan(Gz, Nz, 0), cheb_plan(Gz, Nz, 1), cheb_plan(Gz, Nz, 2)
    D = jacobi.differentiation_matrix(Nz, -0.5, -0.5).toarray()
    dv = np.zeros(Nz)
    dv[:Nz - 1] = np.diagonal(D, 1)
    dvec = dev.from_host(dv)
    cb, gb = c.numel() * 8, g.numel() * 8
    cases = [
        ("z cheb backward plain", cb + gb, lambda: libhip.call("ddh_cheb_backward", p0, ptr(c), ptr(g), nc, inner, dev.stream)),
        ("z cheb backward dual", cb + 2 * gb, lambda: libhip.call("ddh_cheb_backward_dual", p1, ptr(c), ptr(g), ptr(g2), ptr(dvec),
                                                                  nc, inner, dev.stream)),
        ("z cheb forward plain", cb + gb, lambda: libhip.call("ddh_cheb_forward", p0, ptr(g), ptr(c), nc, inner, dev.stream)),
        ("z cheb forward conv(3 bands)", cb + gb, lambda: libhip.call("ddh_cheb_forward", p2, ptr(g), ptr(c), nc, inner, dev.stream)),
    ]
    for name, nbytes, fn in cases:
        ms = timeit(dev, fn) * 1e3
        out[name] = ms
        print("%-34s %7.3f ms  %6.0f GB/s" % (name, ms, nbytes / ms / 1e6), flush=True)
    del c, g, g2
    # ---- x axis: [comp * z grid][kx][ky]
    if os.environ.get("BENCH_X", "1") != "0":
        outer, inner = nc * Gz, Ny
        c = t.randn((outer, Nx, inner), dtype=t.float64, device=dev.tdev)
        g = dev.empty((outer, Gx, inner))
        g2 = dev.empty((outer, Gx, inner))
        pr = plan("ddh_plan_rfft", Gx, Nx)
        cb, gb = c.numel() * 8, g.numel() * 8
        cases = [
            ("x rfft backward", cb + gb, lambda: libhip.call("ddh_rfft_backward", pr, ptr(c), ptr(g), outer, inner, dev.stream)),
            ("x rfft backward dual", cb + 2 * gb, lambda: libhip.call("ddh_rfft_backward_dual", pr, ptr(c), ptr(g), ptr(g2), outer, inner,
                                                                      C.c_double(0.7), dev.stream)),
            ("x rfft forward", cb + gb, lambda: libhip.call("ddh_rfft_forward", pr, ptr(g), ptr(c), outer, inner, dev.stream)),
        ]
        for name, nbytes, fn in cases:
            ms = timeit(dev, fn) * 1e3
            out[name] = ms
            print("%-34s %7.3f ms  %6.0f GB/s" % (name, ms, nbytes / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
