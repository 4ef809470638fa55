"""
TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the UNMODIFIED reference
(through oracle/refshim) on seeded inputs.  Works only where /root/reference exists; the
fixtures it writes are committed so the GPU box (no reference there) can use them.

    python oracle/make_golden.py [transforms] [matrices] [ivp]
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import refshim  # noqa: E402


def golden_transforms():
    d3 = refshim.load_reference()
    from dedalus.core import transforms as T
    rng = np.random.default_rng(1234)
    out = {}
    # --- RealFourier: fast (FFTWRealFFT) and matrix (RealFourierMMT), tests/test_transforms.py:18-57
    cases = []
    for (N, M) in [(24, 16), (16, 16), (8, 16), (30, 20), (12, 8)]:
        for axis, shape in [(0, (N, 6)), (1, (3, N, 4)), (2, (2, 3, N))]:
            g = rng.standard_normal(shape)
            cshape = list(shape)
            cshape[axis] = M
            plan = T.FFTWRealFFT(N, M)
            mmt = T.RealFourierMMT(N, M)
            c = np.zeros(cshape)
            plan.forward(g.copy(), c, axis)
            c_m = np.zeros(cshape)
            mmt.forward(g.copy(), c_m, axis)
            cin = rng.standard_normal(cshape)
            gb = np.zeros(shape)
            plan.backward(cin.copy(), gb, axis)
            gb_m = np.zeros(shape)
            mmt.backward(cin.copy(), gb_m, axis)
            key = "rf_%d_%d_%d" % (N, M, axis)
            cases.append(key)
            out[key + "_g"] = g
            out[key + "_c"] = c
            out[key + "_c_mmt"] = c_m
            out[key + "_cin"] = cin
            out[key + "_gb"] = gb
            out[key + "_gb_mmt"] = gb_m
    out["rf_cases"] = np.array(cases)
    # --- ComplexFourier
    cases = []
    for (N, M) in [(24, 16), (16, 16), (8, 16), (15, 10)]:
        for axis, shape in [(0, (N, 5)), (1, (3, N))]:
            g = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
            cshape = list(shape)
            cshape[axis] = M
            plan = T.FFTWComplexFFT(N, M)
            c = np.zeros(cshape, dtype=complex)
            plan.forward(g.copy(), c, axis)
            cin = rng.standard_normal(cshape) + 1j * rng.standard_normal(cshape)
            gb = np.zeros(shape, dtype=complex)
            plan.backward(cin.copy(), gb, axis)
            key = "cf_%d_%d_%d" % (N, M, axis)
            cases.append(key)
            out[key + "_g"], out[key + "_c"], out[key + "_cin"], out[key + "_gb"] = g, c, cin, gb
    out["cf_cases"] = np.array(cases)
    # --- Chebyshev incl. ultraspherical output, tests/test_transforms.py:117-158
    cases = []
    for alpha in (0, 1, 2):
        for (N, M) in [(24, 16), (16, 16), (8, 16), (18, 12), (15, 15)]:
            for axis, shape in [(0, (N, 6)), (1, (3, N, 4)), (2, (2, 3, N))]:
                a = b = alpha - 0.5
                plan = T.FFTWFastChebyshevTransform(N, M, a, b, -0.5, -0.5)
                mmt = T.JacobiMMT(N, M, a, b, -0.5, -0.5)
                g = rng.standard_normal(shape)
                cshape = list(shape)
                cshape[axis] = M
                c = np.zeros(cshape)
                plan.forward(g.copy(), c, axis)
                c_m = np.zeros(cshape)
                mmt.forward(g.copy(), c_m, axis)
                cin = rng.standard_normal(cshape)
                gb = np.zeros(shape)
                plan.backward(cin.copy(), gb, axis)
                gb_m = np.zeros(shape)
                mmt.backward(cin.copy(), gb_m, axis)
                key = "ch_%d_%d_%d_%d" % (alpha, N, M, axis)
                cases.append(key)
                out[key + "_g"], out[key + "_c"], out[key + "_c_mmt"] = g, c, c_m
                out[key + "_cin"], out[key + "_gb"], out[key + "_gb_mmt"] = cin, gb, gb_m
    out["ch_cases"] = np.array(cases)
    np.savez_compressed(os.path.join(GOLD, "transforms.npz"), **out)
    print("wrote transforms.npz with", len(out), "arrays")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["transforms", "matrices", "ivp"]
    for w in what:
        fn = globals().get("golden_" + w)
        if fn is None:
            print("unknown golden set:", w)
        else:
            fn()
