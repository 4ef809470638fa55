"""CPU check of the second-generation wave-per-line grid stage (dedalus_amd/csrc/ddh_gridwave2.h): the lane code of
the GPU kernel is compiled with g++ (tests/host_emu/emu_gridwave2.cpp: 64 threads = 64 lanes, LDS = a shared array) and
compared with numpy real FFTs in the reference's coefficient convention (cos / -sin pairs, core/transforms.py:469-565).
Index maps, exchanges, the mirror-pair pre- / post-processing and the derivative at load are thereby verified without a GPU;
the GPU tests re-check the compiled kernel."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu") / "libemu_gridwave2.so")
    src = os.path.join(HERE, "host_emu", "emu_gridwave2.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", out, src], check=True)
    lib = C.CDLL(out)
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    lib.emu_gw2_backward.argtypes = [i, i, i, d, vp, vp, vp]
    lib.emu_gw2_forward.argtypes = [i, i, i, i, vp, vp, vp]
    return lib


def dp(a):
    return a.ctypes.data_as(C.c_void_p)


def plan_table(N):
    q = np.arange(N, dtype=np.longdouble)
    a = -2 * np.pi * q / N
    return np.ascontiguousarray(np.stack([np.cos(a), np.sin(a)], -1).astype(np.float64))


def c2r(pairs, N, dscale=0.0):
    """x_n = c_0 + sum_{k>0} (c_k cos(2 pi k n / N) - m_k sin(2 pi k n / N)) for (cos, msin) pairs (c_k, m_k); with
    dscale: the derivative i kappa X, kappa = dscale k."""
    K = len(pairs) - 1
    X = np.zeros(N // 2 + 1, dtype=complex)
    X[:K + 1] = pairs[:, 0] + 1j * pairs[:, 1]
    X[0] = pairs[0, 0]
    if dscale:
        X = 1j * dscale * np.arange(N // 2 + 1) * X
        X[0] = 0.0
    X[1:] *= 0.5
    return np.fft.irfft(X, N) * N


def r2c(x, M):
    N = len(x)
    Y = np.fft.rfft(x)
    out = np.zeros((M // 2, 2))
    out[:, 0] = 2.0 / N * Y.real[:M // 2]
    out[:, 1] = 2.0 / N * Y.imag[:M // 2]
    out[0] = (Y[0].real / N, 0.0)
    return out


@pytest.mark.parametrize("C_", [6, 3])
@pytest.mark.parametrize("twreg", [1, 0, 3])                   # 3: register twiddles + pairs through the LDS staging area
@pytest.mark.parametrize("dscale", [0.0, 1.7])
def test_backward_line(emu, C_, twreg, dscale):
    N = 128 * C_
    M = 2 * (N // 3)
    K = M // 2 - 1
    rng = np.random.default_rng(3 + C_)
    pairs = rng.standard_normal((M // 2, 2))
    pairs[0, 1] = 0.0
    grid = np.full(N, np.nan)
    tw = plan_table(N)
    assert emu.emu_gw2_backward(C_, twreg, K, dscale, dp(pairs), dp(tw), dp(grid)) == 0
    ref = 2.0 * c2r(pairs, N, dscale)
    assert np.abs(grid - ref).max() <= 2e-13 * np.abs(ref).max()


@pytest.mark.parametrize("C_", [6, 3])
@pytest.mark.parametrize("twreg", [1, 0])
def test_forward_line(emu, C_, twreg):
    N = 128 * C_
    M = 2 * (N // 3)
    rng = np.random.default_rng(5 + C_)
    x = rng.standard_normal(N)
    tw = plan_table(N)
    for K in (M // 2 - 1, M // 2 - 7):
        line = np.full((M // 2, 2), np.nan)
        assert emu.emu_gw2_forward(C_, twreg, K, M, dp(x), dp(tw), dp(line)) == 0
        ref = r2c(x, M)
        ref[K + 1:] = 0.0
        assert np.abs(line - ref).max() <= 2e-14 * np.abs(ref).max()


def test_truncated_spectrum_reads_nothing_beyond_K(emu):
    """pairs beyond K are garbage in memory: the loads must not see them (the device uses the buffer range check)"""
    C_, N = 6, 768
    M, K = 512, 200
    rng = np.random.default_rng(9)
    pairs = rng.standard_normal((M // 2, 2))
    pairs[0, 1] = 0.0
    dirty = pairs.copy()
    dirty[K + 1:] = 1e300
    grid = np.zeros(N)
    assert emu.emu_gw2_backward(C_, 1, K, 0.0, dp(dirty), dp(plan_table(N)), dp(grid)) == 0
    ref = 2.0 * c2r(pairs[:K + 1], N)
    assert np.abs(grid - ref).max() <= 2e-13 * np.abs(ref).max()
