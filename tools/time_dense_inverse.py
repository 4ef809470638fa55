"""Timing of ddh_dense_inverse_compute: python tools/time_dense_inverse.py <n> <nsys> <complex 0/1>"""
import sys
import time

import numpy as np
import torch

from dedalus_amd.device import Device
from dedalus_amd.executor import HipExecutor


def main():
    n, nsys, cx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(0)
    dt = np.complex128 if cx else np.float64
    Ms = [(rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cx else 0)).astype(dt) + n * np.eye(n) for _ in range(nsys)]
    Ls = [(rng.standard_normal((n, n))).astype(dt) for _ in range(nsys)]
    ones = [np.ones(n, dtype=np.uint8)] * nsys
    ex = HipExecutor(Device.get())
    di = ex.make_dense_inverse(Ms, Ls, ones, ones, complex_=bool(cx))
    di.compute(1.0, 0.1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = di.compute(1.0, 0.2)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    inv = ex.download(out).view(dt).reshape(nsys, n, n)[0]
    err = np.abs(inv @ (Ms[0] + 0.2 * Ls[0]) - np.eye(n)).max()
    print("n %d nsys %d complex %d: %.2f ms   |inv A - I| %.1e" % (n, nsys, cx, 1e3 * (t1 - t0), err))


if __name__ == "__main__":
    main()
