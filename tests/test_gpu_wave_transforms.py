"""GPU parity of the wave-per-four-pairs strided Chebyshev kernels (csrc/ddh_fftwave.hip) through the C ABI, against the
numpy oracle (reference: core/transforms.py:715-902): whole / partial tiles, several outer indices, enough tiles that
every wave of several workgroups walks more than one tile (the prefetch path), alpha = 0, 1, 2 and the dual backward."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def dev():
    from dedalus_amd.device import Device
    return Device.get()


def _plan(N, M, alpha):
    from dedalus_amd import libhip
    from dedalus_amd.tools import jacobi
    h = C.c_uint64(0)
    if alpha == 0:
        libhip.call("ddh_plan_cheb", C.byref(h), N, M, 0, None, None)
        return h, None
    conv = jacobi.conversion_matrix(M, -0.5, -0.5, alpha - 0.5, alpha - 0.5)
    dense = conv.toarray()
    offs = np.array([o for o in range(M) if np.any(np.diagonal(dense, o) != 0)], dtype=np.int32)
    bands = np.zeros((len(offs), M))
    for d, o in enumerate(offs):
        bands[d, :M - o] = np.diagonal(dense, o)
    libhip.call("ddh_plan_cheb", C.byref(h), N, M, len(offs), libhip.as_ip(offs), libhip.as_dp(bands))
    return h, conv


# every (grid, coefficient) size the wave Chebyshev kernels are instantiated for (csrc/ddh_fftwave.hip: DDH_CHEB_WAVE_SIZES)
WAVE_CHEB_SIZES = [(384, 256), (192, 128), (256, 256), (192, 192), (128, 128), (64, 64), (256, 128), (128, 64)]
WAVE_RFFT_SIZES = [(768, 512), (576, 384), (384, 256), (192, 128)]


def _wave_launch_count(fn):
    """number of launches the library routed to a wave kernel while fn() ran (ddh_fft_wave_launches: a size that silently
    fell back to the workgroup-per-tile kernel would pass the numerical comparison too)"""
    from dedalus_amd import libhip
    n0 = C.c_long(0)
    libhip.call("ddh_fft_wave_launches", C.byref(n0))
    fn()
    n1 = C.c_long(0)
    libhip.call("ddh_fft_wave_launches", C.byref(n1))
    return n1.value - n0.value


@pytest.mark.parametrize("N,M", WAVE_CHEB_SIZES)
@pytest.mark.parametrize("outer,inner", [(1, 2), (2, 10), (3, 64), (1, 8 * 8 * 4 * 2 * 3 + 6), (2, 4096)])
def test_wave_chebyshev_all_paths(dev, N, M, outer, inner):
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    from dedalus_amd.tools import jacobi
    from oracle import np_transforms as npt
    if (N, M) != (384, 256) and (outer, inner) not in ((2, 10), (1, 8 * 8 * 4 * 2 * 3 + 6)):
        pytest.skip("the other tile shapes are checked at the benchmark's size")
    rng = np.random.default_rng(outer * 1000 + inner)
    cs, gs = (outer, M, inner), (outer, N, inner)
    cin = rng.standard_normal(cs) / (1.0 + np.arange(M).reshape(1, -1, 1)) ** 2
    gin = rng.standard_normal(gs)
    d_c, d_g = dev.from_host(cin), dev.from_host(gin)
    D = jacobi.differentiation_matrix(M, -0.5, -0.5).toarray() * (2.0 / 1.7)
    dvec = np.zeros(M)
    dvec[:M - 1] = np.diagonal(D, 1)
    dc = np.ascontiguousarray(np.moveaxis(np.tensordot(D, np.moveaxis(cin, 1, 0), axes=(1, 0)), 0, 1))
    d_v = dev.from_host(dvec)
    plans = {a: _plan(N, M, a) for a in (0, 1, 2)}
    for alpha in (0, 1, 2):
        h, conv = plans[alpha]
        out = dev.empty(cs)
        out.fill_(float("nan"))
        libhip.call("ddh_cheb_forward", h, ptr(d_g), ptr(out), outer, inner, dev.stream)
        dev.sync()
        assert rel(dev.to_host(out), npt.cheb_forward(gin, 1, M, conv)) < 1e-12, ("forward", alpha)
        g = dev.empty(gs)
        g.fill_(float("nan"))
        libhip.call("ddh_cheb_backward", h, ptr(d_c), ptr(g), outer, inner, dev.stream)
        dev.sync()
        assert rel(dev.to_host(g), npt.cheb_backward(cin, 1, N, conv)) < 1e-11, ("backward", alpha)
    ga, gb = dev.empty(gs), dev.empty(gs)
    ga.fill_(float("nan"))
    gb.fill_(float("nan"))
    nw = _wave_launch_count(lambda: libhip.call("ddh_cheb_backward_dual", plans[1][0], ptr(d_c), ptr(ga), ptr(gb), ptr(d_v), outer, inner, dev.stream))
    dev.sync()
    assert nw == 1, "the dual transform of %d <- %d did not run on the wave kernel" % (N, M)
    assert np.array_equal(dev.to_host(d_c), cin)
    assert rel(dev.to_host(ga), npt.cheb_backward(cin, 1, N, None)) < 1e-12
    assert rel(dev.to_host(gb), npt.cheb_backward(dc, 1, N, plans[1][1])) < 1e-11


@pytest.mark.parametrize("N,M", WAVE_CHEB_SIZES)
@pytest.mark.parametrize("nlines", [2, 8, 18, 8 * 4 * 4 * 7 + 6, 3 * 256 * 128])
def test_wave_chebyshev_along_the_contiguous_axis(dev, N, M, nlines):
    """The shell's radial transforms (contiguous lines, 192 <- 128) and the vertical axis of 2-D problems (384 <- 256;
    wave_cheb_contig_kernel): one pair, one tile, a partial tile, several tiles per wave with a ragged end, and the shell
    configuration's own line count; alpha = 0, 1, 2 both ways against the numpy oracle (reference:
    core/transforms.py:715-902), the round trip, and the dual backward transform (field + z derivative)."""
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    from dedalus_amd.tools import jacobi
    from oracle import np_transforms as npt
    if (N, M) not in ((192, 128), (384, 256)) and nlines != 8 * 4 * 4 * 7 + 6:
        pytest.skip("the other line counts are checked at the configurations' sizes")
    if (N, M) == (384, 256) and nlines > 10000:
        nlines = 2 * 512 * 3                 # (2-D Rayleigh-Benard 512 x 256: 3 components of 512 contiguous lines)
    big = nlines > 10000
    rng = np.random.default_rng(nlines)
    cin = rng.standard_normal((nlines, M)) / (1.0 + np.arange(M).reshape(1, -1)) ** 2
    gin = rng.standard_normal((nlines, N))
    d_c, d_g = dev.from_host(cin), dev.from_host(gin)
    sub = slice(None) if not big else slice(0, nlines, 997)
    for alpha in (0, 1, 2):
        h, conv = _plan(N, M, alpha)
        out = dev.empty((nlines, M))
        out.fill_(float("nan"))
        libhip.call("ddh_cheb_forward", h, ptr(d_g), ptr(out), nlines, 1, dev.stream)
        dev.sync()
        o = dev.to_host(out)
        assert np.isfinite(o).all()
        assert rel(o[sub], npt.cheb_forward(gin[sub], 1, M, conv)) < 1e-12, ("forward", alpha)
        g = dev.empty((nlines, N))
        g.fill_(float("nan"))
        libhip.call("ddh_cheb_backward", h, ptr(d_c), ptr(g), nlines, 1, dev.stream)
        dev.sync()
        gh = dev.to_host(g)
        assert np.isfinite(gh).all()
        assert rel(gh[sub], npt.cheb_backward(cin[sub], 1, N, conv)) < 1e-11, ("backward", alpha)
        assert np.array_equal(dev.to_host(d_c), cin) and np.array_equal(dev.to_host(d_g), gin)
        c2 = dev.empty((nlines, M))
        libhip.call("ddh_cheb_forward", h, ptr(g), ptr(c2), nlines, 1, dev.stream)
        dev.sync()
        assert rel(dev.to_host(c2), cin) < 1e-12, ("round trip", alpha)
        if alpha == 1:
            D = jacobi.differentiation_matrix(M, -0.5, -0.5).toarray() * (2.0 / 1.7)
            dvec = np.zeros(M)
            dvec[:M - 1] = np.diagonal(D, 1)
            d_v = dev.from_host(dvec)
            ga, gb = dev.empty((nlines, N)), dev.empty((nlines, N))
            ga.fill_(float("nan"))
            gb.fill_(float("nan"))
            nw = _wave_launch_count(lambda: libhip.call("ddh_cheb_backward_dual", h, ptr(d_c), ptr(ga), ptr(gb), ptr(d_v), nlines, 1, dev.stream))
            dev.sync()
            assert nw == 1, "the contiguous dual transform of %d <- %d did not run on the wave kernel" % (N, M)
            h0, _ = _plan(N, M, 0)
            g0 = dev.empty((nlines, N))
            libhip.call("ddh_cheb_backward", h0, ptr(d_c), ptr(g0), nlines, 1, dev.stream)
            dev.sync()
            assert np.array_equal(dev.to_host(ga), dev.to_host(g0))
            assert rel(dev.to_host(gb)[sub], npt.cheb_backward(cin[sub] @ D.T, 1, N, conv)) < 1e-11


def test_wave_chebyshev_round_trip_full_size(dev):
    """512 x 512 x 256 shape (one component): forward(backward(c)) == c, and the dual's derivative output equals the
    backward transform of the differentiated coefficients taken by the plain conversion path."""
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    from dedalus_amd.tools import jacobi
    t = dev.torch
    N, M, inner = 384, 256, 512 * 512
    gen = t.Generator(device=dev.tdev)
    gen.manual_seed(11)
    c = t.randn((1, M, inner), dtype=t.float64, device=dev.tdev, generator=gen)
    c /= (1.0 + t.arange(M, dtype=t.float64, device=dev.tdev).reshape(1, -1, 1)) ** 2
    g, g2, c2 = dev.empty((1, N, inner)), dev.empty((1, N, inner)), dev.empty((1, M, inner))
    h0, _ = _plan(N, M, 0)
    h1, _ = _plan(N, M, 1)
    libhip.call("ddh_cheb_backward", h0, ptr(c), ptr(g), 1, inner, dev.stream)
    libhip.call("ddh_cheb_forward", h0, ptr(g), ptr(c2), 1, inner, dev.stream)
    dev.sync()
    assert float((c2 - c).norm() / c.norm()) < 1e-13
    D = jacobi.differentiation_matrix(M, -0.5, -0.5).toarray()
    dv = np.zeros(M)
    dv[:M - 1] = np.diagonal(D, 1)
    d_v = dev.from_host(dv)
    ga, gb = dev.empty((1, N, inner)), dev.empty((1, N, inner))
    libhip.call("ddh_cheb_backward_dual", h1, ptr(c), ptr(ga), ptr(gb), ptr(d_v), 1, inner, dev.stream)
    dc = t.zeros_like(c)
    dc[:, :M - 1] = c[:, 1:] * d_v[:M - 1].reshape(1, -1, 1)
    libhip.call("ddh_cheb_backward", h1, ptr(dc), ptr(g2), 1, inner, dev.stream)
    dev.sync()
    assert t.equal(ga, g)
    assert float((gb - g2).norm() / g2.norm()) < 1e-13


@pytest.mark.parametrize("N,M", WAVE_RFFT_SIZES)
@pytest.mark.parametrize("outer,inner", [(1, 2), (2, 10), (3, 64), (1, 8 * 8 * 4 * 2 + 6), (2, 1024)])
def test_wave_real_fourier_all_paths(dev, N, M, outer, inner):
    """wave-per-four-pairs real FFT with 3/2 dealiasing (three length-N/3 transforms): backward, differentiated
    backward, dual backward (bitwise equal to the two single launches) and forward, vs the oracle
    (core/transforms.py:469-565)."""
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    from oracle import np_transforms as npt
    rng = np.random.default_rng(N + outer * 1000 + inner)
    cs, gs = (outer, M, inner), (outer, N, inner)
    cin = rng.standard_normal(cs)
    gin = rng.standard_normal(gs)
    h = C.c_uint64(0)
    libhip.call("ddh_plan_rfft", C.byref(h), N, M)
    d_c, d_g = dev.from_host(cin), dev.from_host(gin)
    dscale = 2 * np.pi / 3.0
    outs = [dev.empty(gs) for _ in range(4)]
    for o in outs:
        o.fill_(float("nan"))
    cf = dev.empty(cs)
    cf.fill_(float("nan"))

    def run():
        libhip.call("ddh_rfft_backward_dual", h, ptr(d_c), ptr(outs[0]), ptr(outs[1]), outer, inner, dscale, dev.stream)
        libhip.call("ddh_rfft_backward", h, ptr(d_c), ptr(outs[2]), outer, inner, dev.stream)
        libhip.call("ddh_rfft_backward_deriv", h, ptr(d_c), ptr(outs[3]), outer, inner, dscale, dev.stream)
        libhip.call("ddh_rfft_forward", h, ptr(d_g), ptr(cf), outer, inner, dev.stream)
    assert _wave_launch_count(run) == 4, "a real-Fourier transform of %d <- %d fell back to the workgroup-per-tile kernel" % (N, M)
    dev.sync()
    g, gd, g1, gd1 = [dev.to_host(o) for o in outs]
    assert np.array_equal(dev.to_host(d_c), cin)
    assert np.array_equal(g, g1) and np.array_equal(gd, gd1)
    k = dscale * np.arange(M // 2)
    dc = np.empty_like(cin)
    dc[:, 0::2] = -k.reshape(1, -1, 1) * cin[:, 1::2]
    dc[:, 1::2] = k.reshape(1, -1, 1) * cin[:, 0::2]
    assert rel(g, npt.rfft_backward(cin, 1, N)) < 1e-12
    assert rel(gd, npt.rfft_backward(dc, 1, N)) < 1e-12
    assert rel(dev.to_host(cf), npt.rfft_forward(gin, 1, M)) < 1e-12


def test_fused_grid_stage_generations():
    """The fused y stage has two kernel generations: gw2::gridwave2_bilinear_kernel (csrc/ddh_gridwave2.hip: C x 8 x 8, LDS-DMA
    operand staging two operands ahead; its lane code is also checked on the CPU, tests/test_host_emu_gridwave2.py) is the
    default for full 3/2-padded lines of 768 / 384 points, gw::gridwave_bilinear_kernel for everything else.  The same
    oracle comparison for every selectable variant, each in a subprocess (the switches are read once per process):
    DDH_GW_V2=0 first generation; =2 second generation also for truncated spectra; DDH_GW_DMA=0 register loads, =2 LDS-DMA
    without register twiddles."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = "test_fused_grid_stage and (768-512 or 384-256 or 768-40)"
    for extra in ({"DDH_GW_V2": "0"}, {"DDH_GW_V2": "2"}, {"DDH_GW_V2": "2", "DDH_GW_DMA": "0"}, {"DDH_GW_DMA": "2"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_transforms.py"), "-q", "-m", "gpu",
                            "-k", sel, "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, (extra, r.stdout[-3000:] + r.stderr[-2000:])
        assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("N,M,B", [(384, 256, 64), (768, 512, 128), (192, 128, 64)])
def test_windows_of_the_blocked_stage_layout_equal_the_whole_transform(dev, N, M, B):
    """ddh_fft_set_stage_window (round 6: the grid stage of a sharded run in windows of z planes): planes z0 .. z0 + cw of
    every component of the blocked stage array [comp][p][gz][B][ky], transformed into / out of an array that holds cw planes
    per component, are bit for bit the planes of the transform over all gz planes -- backward, dual and forward."""
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    ncomp, gz, ky, cw = 2, 6, 16, 2
    rng = np.random.default_rng(N + B)
    h = C.c_uint64(0)
    libhip.call("ddh_plan_rfft", C.byref(h), N, M)
    libhip.call("ddh_fft_set_stage_layout", h, gz)
    libhip.call("ddh_fft_set_stage_block", h, B)
    stage = dev.from_host(rng.standard_normal((ncomp, M // B, gz, B, ky)))         # [comp][p][gz][B][ky]
    full, full_d = dev.empty((ncomp, gz, N, ky)), dev.empty((ncomp, gz, N, ky))
    libhip.call("ddh_rfft_backward_dual", h, ptr(stage), ptr(full), ptr(full_d), ncomp * gz, ky, 1.7, dev.stream)
    grid = dev.from_host(rng.standard_normal((ncomp, gz, N, ky)))
    back = dev.empty((ncomp, M // B, gz, B, ky))
    libhip.call("ddh_rfft_forward", h, ptr(grid), ptr(back), ncomp * gz, ky, dev.stream)
    win_back = dev.empty((ncomp, M // B, gz, B, ky))
    win_back.fill_(float("nan"))
    for z0 in range(0, gz, cw):
        libhip.call("ddh_fft_set_stage_window", h, z0, cw)
        w, wd, w1 = dev.empty((ncomp, cw, N, ky)), dev.empty((ncomp, cw, N, ky)), dev.empty((ncomp, cw, N, ky))
        libhip.call("ddh_rfft_backward_dual", h, ptr(stage), ptr(w), ptr(wd), ncomp * cw, ky, 1.7, dev.stream)
        libhip.call("ddh_rfft_backward", h, ptr(stage), ptr(w1), ncomp * cw, ky, dev.stream)
        assert (w == full[:, z0:z0 + cw]).all() and (wd == full_d[:, z0:z0 + cw]).all() and (w1 == w).all()
        gw = grid[:, z0:z0 + cw].contiguous()
        libhip.call("ddh_rfft_forward", h, ptr(gw), ptr(win_back), ncomp * cw, ky, dev.stream)
    libhip.call("ddh_fft_set_stage_window", h, 0, 0)
    dev.sync()
    assert (win_back == back).all()
    libhip.call("ddh_destroy", h)
