"""CPU check of the wave-level strided transforms (dedalus_amd/csrc/ddh_wavefft.h): the lane code of the GPU kernels is
compiled with g++ (tests/host_emu/emu_wavefft.cpp: 64 threads = 64 lanes, LDS = a shared array) and compared with numpy
FFTs and with the oracle's Chebyshev transforms (reference: core/transforms.py:715-902).  Index maps, exchanges, the
conversion solve and the arithmetic are thereby verified without a GPU; the GPU tests re-check the compiled kernels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import np_transforms as npt
from dedalus_amd.tools import jacobi

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu") / "libemu_wavefft.so")
    src = os.path.join(HERE, "host_emu", "emu_wavefft.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", out, src], check=True)
    lib = C.CDLL(out)
    vp, i, l = C.c_void_p, C.c_int, C.c_long
    lib.emu_fft4.argtypes = [i, i, vp, vp, vp]
    lib.emu_cheb_bwd.argtypes = [i, i, i, vp, vp, vp, vp, i, vp, vp, vp, l, l]
    lib.emu_cheb_fwd.argtypes = [i, i, vp, vp, i, vp, vp, vp, vp, l, l]
    lib.emu_cheb_bwd_contig.argtypes = [i, i, i, vp, vp, vp, i, vp, vp, l, vp, vp]
    lib.emu_cheb_fwd_contig.argtypes = [i, i, vp, vp, i, vp, vp, vp, vp, l]
    lib.emu_rfft_bwd.argtypes = [i, i, C.c_double, C.c_double, vp, vp, vp, vp, l, l]
    lib.emu_rfft_fwd.argtypes = [i, vp, vp, vp, l, l]
    lib.emu_rfft_blocked.argtypes = [i, i, l, vp, vp, vp, vp, l]
    return lib


def dp(a):
    return a.ctypes.data_as(C.c_void_p)


def tables(N):
    q = np.arange(N, dtype=np.longdouble)
    a = -2 * np.pi * q / N
    h = -np.longdouble(np.pi) * q / (2 * N)
    tw = np.ascontiguousarray(np.stack([np.cos(a), np.sin(a)], -1).astype(np.float64))
    half = np.ascontiguousarray(np.stack([np.cos(h), np.sin(h)], -1).astype(np.float64))
    return tw, half


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def conv_bands(M, alpha):
    conv = jacobi.conversion_matrix(M, -0.5, -0.5, alpha - 0.5, alpha - 0.5)
    dense = conv.toarray()
    offs = np.array([o for o in range(M) if np.any(np.diagonal(dense, o) != 0)], dtype=np.int32)
    bands = np.zeros((len(offs), M))
    for d, o in enumerate(offs):
        bands[d, :M - o] = np.diagonal(dense, o)
    return conv, offs, bands


@pytest.mark.parametrize("N", [384, 256, 192, 128, 64])
@pytest.mark.parametrize("sign", [-1, 1])
def test_wave_fft_of_four_interleaved_lines(emu, N, sign):
    rng = np.random.default_rng(N + sign)
    tw, _ = tables(N)
    x = rng.standard_normal((N, 4, 2))
    X = np.zeros_like(x)
    assert emu.emu_fft4(N, sign, dp(tw), dp(x), dp(X)) == 0
    xc = x[..., 0] + 1j * x[..., 1]
    ref = np.fft.fft(xc, axis=0) if sign < 0 else np.fft.ifft(xc, axis=0) * N
    assert rel(X[..., 0] + 1j * X[..., 1], ref) < 1e-14


WAVE_CHEB_SIZES = [(384, 256), (192, 128), (256, 256), (192, 192), (128, 128), (64, 64), (256, 128), (128, 64)]


@pytest.mark.parametrize("NM", WAVE_CHEB_SIZES)
@pytest.mark.parametrize("oi", [(2, 12), (1, 2), (3, 10)])
def test_wave_chebyshev_matches_the_oracle(emu, NM, oi):
    """whole and partial tiles (inner = 10: five pairs = one full tile + one pair), several outer indices; every
    instantiated (grid, coefficient) size of csrc/ddh_fftwave.hip (DDH_CHEB_WAVE_SIZES)"""
    N, M = NM
    if NM != (384, 256) and oi != (3, 10):
        pytest.skip("the other tile shapes are checked at the benchmark's size")
    outer, inner = oi
    shape = (outer, M, inner)
    rng = np.random.default_rng(inner)
    tw, half = tables(N)
    zero = np.zeros(3 * M)
    cin = rng.standard_normal(shape) / (1.0 + np.arange(M).reshape(1, -1, 1)) ** 2
    gs = (outer, N, inner)
    g = np.full(gs, np.nan)
    assert emu.emu_cheb_bwd(N, M, 0, dp(tw), dp(half), dp(zero), dp(zero), 1, dp(cin), dp(g), None, outer, inner) == 0
    assert rel(g, npt.cheb_backward(cin, 1, N, None)) < 1e-14
    conv, offs, bands = conv_bands(M, 1)
    assert list(offs) == [0, 2]
    bsub = np.zeros((2, M))
    bsub[0] = 1.0 / bands[0]
    bsub[1, :M - 2] = bands[1, :M - 2] / bands[0, :M - 2]
    g2 = np.full(gs, np.nan)
    emu.emu_cheb_bwd(N, M, 2, dp(tw), dp(half), dp(bsub), dp(zero), 2, dp(cin), dp(g2), None, outer, inner)
    assert rel(g2, npt.cheb_backward(cin, 1, N, conv)) < 1e-13
    D = jacobi.differentiation_matrix(M, -0.5, -0.5).toarray() * (2.0 / 1.7)
    dvec = np.zeros(M)
    dvec[:M - 1] = np.diagonal(D, 1)
    dc = np.ascontiguousarray(np.moveaxis(np.tensordot(D, np.moveaxis(cin, 1, 0), axes=(1, 0)), 0, 1))
    ga, gb = np.full(gs, np.nan), np.full(gs, np.nan)
    emu.emu_cheb_bwd(N, M, 1, dp(tw), dp(half), dp(bsub), dp(dvec), 2, dp(cin), dp(ga), dp(gb), outer, inner)
    assert np.array_equal(ga, g)                          # the dual's first pass IS the plain pass
    assert rel(gb, npt.cheb_backward(dc, 1, N, conv)) < 1e-13
    gin = rng.standard_normal(gs)
    for alpha in (0, 1, 2):
        out = np.full(shape, np.nan)
        if alpha == 0:
            emu.emu_cheb_fwd(N, M, dp(tw), dp(half), 0, None, None, dp(gin), dp(out), outer, inner)
            cv = None
        else:
            cv, offs, bands = conv_bands(M, alpha)
            emu.emu_cheb_fwd(N, M, dp(tw), dp(half), len(offs), dp(offs), dp(bands), dp(gin), dp(out), outer, inner)
        assert rel(out, npt.cheb_forward(gin, 1, M, cv)) < 1e-14


@pytest.mark.parametrize("NM", WAVE_CHEB_SIZES)
@pytest.mark.parametrize("nlines", [8, 18, 2])
def test_wave_chebyshev_along_the_contiguous_axis(emu, NM, nlines):
    """wave_cheb_contig_kernel (the shell's radial transforms, 192 <- 128; the vertical axis of 2-D Rayleigh-Benard,
    384 <- 256): 8 contiguous lines per wave, grid lines staged line-major in LDS; whole tiles, a partial tile (18 = 2
    tiles + one pair), a single pair; plain, conversion solve (ultraspherical alpha = 1, 2), dual (field + z derivative),
    forward with conversion bands"""
    N, M = NM
    if NM not in ((192, 128), (384, 256)) and nlines != 18:
        pytest.skip("the other line counts are checked at the configurations' sizes")
    rng = np.random.default_rng(nlines)
    tw, half = tables(N)
    zero = np.zeros(3 * M)
    cin = rng.standard_normal((nlines, M)) / (1.0 + np.arange(M).reshape(1, -1)) ** 2
    g = np.full((nlines, N), np.nan)
    assert emu.emu_cheb_bwd_contig(N, M, 0, dp(tw), dp(half), dp(zero), 1, dp(cin), dp(g), nlines, None, None) == 0
    assert rel(g, npt.cheb_backward(cin, 1, N, None)) < 1e-14
    for alpha in (1, 2):
        conv, offs, bands = conv_bands(M, alpha)
        if len(offs) != 2:
            continue                                     # (first-order back substitution only, as on the device)
        go = int(offs[1])
        bsub = np.zeros((2, M))
        bsub[0] = 1.0 / bands[0]
        bsub[1, :M - go] = bands[1, :M - go] / bands[0, :M - go]
        g2 = np.full((nlines, N), np.nan)
        assert emu.emu_cheb_bwd_contig(N, M, 2, dp(tw), dp(half), dp(bsub), go, dp(cin), dp(g2), nlines, None, None) == 0
        assert rel(g2, npt.cheb_backward(cin, 1, N, conv)) < 1e-13
        if alpha == 1:
            # dual: the plain transform and the z derivative (one superdiagonal into the alpha = 1 basis) from one read
            D = jacobi.differentiation_matrix(M, -0.5, -0.5).toarray() * (2.0 / 1.7)
            dvec = np.zeros(M)
            dvec[:M - 1] = np.diagonal(D, 1)
            ga, gb = np.full((nlines, N), np.nan), np.full((nlines, N), np.nan)
            assert emu.emu_cheb_bwd_contig(N, M, 1, dp(tw), dp(half), dp(bsub), go, dp(cin), dp(ga), nlines, dp(dvec), dp(gb)) == 0
            assert np.array_equal(ga, g)
            assert rel(gb, npt.cheb_backward(cin @ D.T, 1, N, conv)) < 1e-13
    gin = rng.standard_normal((nlines, N))
    for alpha in (0, 1, 2):
        out = np.full((nlines, M), np.nan)
        if alpha == 0:
            emu.emu_cheb_fwd_contig(N, M, dp(tw), dp(half), 0, None, None, dp(gin), dp(out), nlines)
            cv = None
        else:
            cv, offs, bands = conv_bands(M, alpha)
            emu.emu_cheb_fwd_contig(N, M, dp(tw), dp(half), len(offs), dp(offs), dp(bands), dp(gin), dp(out), nlines)
        assert rel(out, npt.cheb_forward(gin, 1, M, cv)) < 1e-14


@pytest.mark.parametrize("N", [768, 576, 384, 192])
@pytest.mark.parametrize("shape_oi", [(2, 10), (1, 8), (3, 2)])
def test_wave_real_fourier_matches_the_oracle(emu, N, shape_oi):
    """3/2-dealiased real FFT along a strided axis as three length-N/3 transforms (core/transforms.py:469-565)"""
    M = 2 * N // 3
    outer, inner = shape_oi
    rng = np.random.default_rng(N + inner)
    tw, _ = tables(N)
    cs, gs = (outer, M, inner), (outer, N, inner)
    cin = rng.standard_normal(cs)
    g, gd, g1, g2 = (np.full(gs, np.nan) for _ in range(4))
    dscale = 2 * np.pi / 4.0
    assert emu.emu_rfft_bwd(N, 1, 0.0, dscale, dp(tw), dp(cin), dp(g), dp(gd), outer, inner) == 0
    emu.emu_rfft_bwd(N, 0, 0.0, 0.0, dp(tw), dp(cin), dp(g1), None, outer, inner)
    emu.emu_rfft_bwd(N, 0, dscale, 0.0, dp(tw), dp(cin), dp(g2), None, outer, inner)
    assert np.array_equal(g, g1) and np.array_equal(gd, g2)
    k = dscale * np.arange(M // 2)
    dc = np.empty_like(cin)
    dc[:, 0::2] = -k.reshape(1, -1, 1) * cin[:, 1::2]
    dc[:, 1::2] = k.reshape(1, -1, 1) * cin[:, 0::2]
    assert rel(g, npt.rfft_backward(cin, 1, N)) < 1e-14
    assert rel(gd, npt.rfft_backward(dc, 1, N)) < 1e-14
    gin = rng.standard_normal(gs)
    out = np.full(cs, np.nan)
    emu.emu_rfft_fwd(N, dp(tw), dp(gin), dp(out), outer, inner)
    assert rel(out, npt.rfft_forward(gin, 1, M)) < 1e-14


@pytest.mark.parametrize("N,B", [(768, 64), (768, 128), (768, 256), (384, 64), (384, 128)])
def test_wave_real_fourier_reads_and_writes_the_blocked_coefficient_layout(emu, N, B):
    """[kx / B][z][kx % B][ky]: B = 64 is the x-blocked stage layout of one rank, B = nx / P the layout the all-to-all of a
    sharded run delivers ([p][z][nx / P][ky], core/transposes.pyx:359-445 would unpack it): the x transforms read and write it
    in place of the natural [z][kx][ky]"""
    M = 2 * N // 3
    gz, inner = 3, 10
    rng = np.random.default_rng(N + B)
    tw, _ = tables(N)
    nat = rng.standard_normal((gz, M, inner))                                  # [z][kx][ky]
    blocked = np.ascontiguousarray(nat.reshape(gz, M // B, B, inner).transpose(1, 0, 2, 3))
    grid = np.full((gz, N, inner), np.nan)
    back = np.full_like(blocked, np.nan)
    assert emu.emu_rfft_blocked(N, B, gz, dp(tw), dp(blocked), dp(grid), dp(back), inner) == 0
    assert rel(grid, npt.rfft_backward(nat, 1, N)) < 1e-14
    want = nat.copy()
    want[:, 1, :] = 0.0                                                        # the msin row of k = 0 is no mode
    want_b = np.ascontiguousarray(want.reshape(gz, M // B, B, inner).transpose(1, 0, 2, 3))
    assert rel(back, want_b) < 1e-13
