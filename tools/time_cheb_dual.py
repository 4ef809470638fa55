"""Kernel timing: ddh_cheb_backward (plain plan, banded plan) vs ddh_cheb_backward_dual on [nc][256 -> 384][512*512]."""
import ctypes as C
import sys
import time

import numpy as np
import torch

from dedalus_amd import libhip
from dedalus_amd.device import Device, ptr
from dedalus_amd.tools import jacobi


def plan(N, M, alpha):
    h = C.c_uint64(0)
    if alpha == 0:
        libhip.call("ddh_plan_cheb", C.byref(h), N, M, 0, None, None)
        return h
    dense = jacobi.conversion_matrix(M, -0.5, -0.5, alpha - 0.5, alpha - 0.5).toarray()
    offs = np.array([o for o in range(M) if np.any(np.diagonal(dense, o) != 0)], dtype=np.int32)
    bands = np.zeros((len(offs), M))
    for d, o in enumerate(offs):
        bands[d, :M - o] = np.diagonal(dense, o)
    libhip.call("ddh_plan_cheb", C.byref(h), N, M, len(offs), libhip.as_ip(offs), libhip.as_dp(bands))
    return h


def main():
    dev = Device.get()
    # argv: number of components, inner extent (lines per outer block; the total line count is kept at nc * 512 * 512)
    N, M, nc = 384, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 3
    inner = int(sys.argv[2]) if len(sys.argv) > 2 else 512 * 512
    nc = nc * (512 * 512 // inner)
    p0, p1 = plan(N, M, 0), plan(N, M, 1)
    c = torch.randn((nc, M, inner), dtype=torch.float64, device="cuda")
    g0 = torch.empty((nc, N, inner), dtype=torch.float64, device="cuda")
    g1 = torch.empty_like(g0)
    dv = torch.arange(1, M + 1, dtype=torch.float64, device="cuda")

    def t(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    s = dev.stream
    print("plain  %.3f ms" % t(lambda: libhip.call("ddh_cheb_backward", p0, ptr(c), ptr(g0), nc, inner, s)))
    print("banded %.3f ms" % t(lambda: libhip.call("ddh_cheb_backward", p1, ptr(c), ptr(g1), nc, inner, s)))
    print("dual   %.3f ms" % t(lambda: libhip.call("ddh_cheb_backward_dual", p1, ptr(c), ptr(g0), ptr(g1), ptr(dv), nc,
                                                    inner, s)))


if __name__ == "__main__":
    main()
