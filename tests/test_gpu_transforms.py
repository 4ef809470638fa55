"""GPU parity: HIP transforms (through the C ABI) vs the reference's own outputs (golden fixtures)
and vs the numpy oracle on seeded inputs.  float64; tolerance rel-L2 <= 1e-12 per transform
(SURVEY.md section 8d)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def dev():
    from dedalus_amd.device import Device
    return Device.get()


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "transforms.npz"))


def _run(dev, fname, plan, src, out_shape, axis):
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    src = np.ascontiguousarray(src)
    is_c = np.iscomplexobj(src)
    d_src = dev.from_host(src)
    d_dst = dev.empty(out_shape, np.complex128 if is_c else np.float64)
    d_dst.fill_(float("nan"))
    outer = int(np.prod(src.shape[:axis]))
    inner = int(np.prod(src.shape[axis + 1:]))
    libhip.call(fname, plan, ptr(d_src), ptr(d_dst), outer, inner, dev.stream)
    dev.sync()
    return dev.to_host(d_dst)


def _plan(name, *args):
    from dedalus_amd import libhip
    h = C.c_uint64(0)
    libhip.call(name, C.byref(h), *args)
    return h


def _cheb_plan(N, M, alpha):
    from dedalus_amd import libhip
    from dedalus_amd.tools import jacobi
    if alpha == 0:
        return _plan("ddh_plan_cheb", N, M, 0, None, None), None
    conv = jacobi.conversion_matrix(M, -0.5, -0.5, alpha - 0.5, alpha - 0.5)
    dense = conv.toarray()
    offs = np.array([o for o in range(M) if np.any(np.diagonal(dense, o) != 0)], dtype=np.int32)
    bands = np.zeros((len(offs), M))
    for d, o in enumerate(offs):
        bands[d, :M - o] = np.diagonal(dense, o)
    return _plan("ddh_plan_cheb", N, M, len(offs), libhip.as_ip(offs), libhip.as_dp(bands)), conv


def _parse(key):
    return [int(x) for x in key.split("_")[1:]]


def test_real_fourier_vs_reference_golden(dev, gold):
    for key in gold["rf_cases"]:
        N, M, axis = _parse(key)
        plan = _plan("ddh_plan_rfft", N, M)
        g, c, cin, gb = (gold[key + s] for s in ("_g", "_c", "_cin", "_gb"))
        assert rel(_run(dev, "ddh_rfft_forward", plan, g, c.shape, axis), c) < TOL, key
        assert rel(_run(dev, "ddh_rfft_backward", plan, cin, gb.shape, axis), gb) < TOL, key
        # and the reference's matrix-multiply definition (tests/test_transforms.py:18-57)
        assert rel(_run(dev, "ddh_rfft_forward", plan, g, c.shape, axis), gold[key + "_c_mmt"]) < TOL, key


def test_complex_fourier_vs_reference_golden(dev, gold):
    for key in gold["cf_cases"]:
        N, M, axis = _parse(key)
        plan = _plan("ddh_plan_cfft", N, M)
        g, c, cin, gb = (gold[key + s] for s in ("_g", "_c", "_cin", "_gb"))
        assert rel(_run(dev, "ddh_cfft_forward", plan, g, c.shape, axis), c) < TOL, key
        assert rel(_run(dev, "ddh_cfft_backward", plan, cin, gb.shape, axis), gb) < TOL, key


def test_chebyshev_vs_reference_golden(dev, gold):
    for key in gold["ch_cases"]:
        alpha, N, M, axis = _parse(key)
        plan, _ = _cheb_plan(N, M, alpha)
        g, c, cin, gb = (gold[key + s] for s in ("_g", "_c", "_cin", "_gb"))
        assert rel(_run(dev, "ddh_cheb_forward", plan, g, c.shape, axis), c) < TOL, key
        assert rel(_run(dev, "ddh_cheb_backward", plan, cin, gb.shape, axis), gb) < 1e-11, key
        assert rel(_run(dev, "ddh_cheb_forward", plan, g, c.shape, axis), gold[key + "_c_mmt"]) < TOL, key


@pytest.mark.parametrize("N,M", [(768, 512), (384, 256), (1536, 1024), (96, 64), (60, 40), (210, 140), (64, 64)])
@pytest.mark.parametrize("shape_kind", ["contig", "strided_even", "strided_odd"])
def test_real_fourier_vs_oracle(dev, N, M, shape_kind):
    from oracle import np_transforms as npt
    rng = np.random.default_rng(1234)
    shape, axis = {"contig": ((7, N), 1), "strided_even": ((3, N, 10), 1), "strided_odd": ((2, N, 7), 1)}[shape_kind]
    plan = _plan("ddh_plan_rfft", N, M)
    g = rng.standard_normal(shape)
    cs = list(shape)
    cs[axis] = M
    c_ref = npt.rfft_forward(g, axis, M)
    assert rel(_run(dev, "ddh_rfft_forward", plan, g, cs, axis), c_ref) < TOL
    cin = rng.standard_normal(cs)
    assert rel(_run(dev, "ddh_rfft_backward", plan, cin, shape, axis), npt.rfft_backward(cin, axis, N)) < TOL


@pytest.mark.parametrize("N,M", [(768, 512), (96, 64), (60, 40), (64, 64)])
@pytest.mark.parametrize("shape_kind", ["contig", "strided_even", "strided_odd"])
def test_real_fourier_dual_backward(dev, N, M, shape_kind):
    """ddh_rfft_backward_dual == ddh_rfft_backward and ddh_rfft_backward_deriv of the same coefficients (bit for bit: the
    passes run the same arithmetic) and both agree with the oracle."""
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    from oracle import np_transforms as npt
    rng = np.random.default_rng(77)
    shape, axis = {"contig": ((7, N), 1), "strided_even": ((3, N, 10), 1), "strided_odd": ((2, N, 7), 1)}[shape_kind]
    plan = _plan("ddh_plan_rfft", N, M)
    cs = list(shape)
    cs[axis] = M
    cin = rng.standard_normal(cs)
    dscale = 2 * np.pi / 4.0
    outer, inner = int(np.prod(shape[:axis])), int(np.prod(shape[axis + 1:]))
    d_c = dev.from_host(cin)
    outs = [dev.empty(shape, np.float64) for _ in range(4)]
    for o in outs:
        o.fill_(float("nan"))
    libhip.call("ddh_rfft_backward_dual", plan, ptr(d_c), ptr(outs[0]), ptr(outs[1]), outer, inner, dscale, dev.stream)
    libhip.call("ddh_rfft_backward", plan, ptr(d_c), ptr(outs[2]), outer, inner, dev.stream)
    libhip.call("ddh_rfft_backward_deriv", plan, ptr(d_c), ptr(outs[3]), outer, inner, dscale, dev.stream)
    dev.sync()
    g, gd, g1, gd1 = [dev.to_host(o) for o in outs]
    assert np.array_equal(dev.to_host(d_c), cin)
    assert np.array_equal(g, g1) and np.array_equal(gd, gd1)
    k = (dscale * np.arange(M // 2))
    sh = [1] * len(shape)
    sh[axis] = -1
    dc = np.empty_like(cin)
    ev = [slice(None)] * len(shape)
    od = list(ev)
    ev[axis], od[axis] = slice(0, None, 2), slice(1, None, 2)
    dc[tuple(ev)] = -k.reshape(sh) * cin[tuple(od)]
    dc[tuple(od)] = k.reshape(sh) * cin[tuple(ev)]
    assert rel(g, npt.rfft_backward(cin, axis, N)) < TOL
    assert rel(gd, npt.rfft_backward(dc, axis, N)) < TOL


@pytest.mark.parametrize("N,M", [(384, 256), (96, 64), (90, 60), (64, 64)])
@pytest.mark.parametrize("shape_kind", ["contig", "strided_even", "strided_odd"])
def test_chebyshev_dual_backward(dev, N, M, shape_kind):
    """ddh_cheb_backward_dual: plain transform + transform of the z-derivative (superdiagonal operator, taken in the
    (a0+1, b0+1) basis) == the two separate paths (plain plan; derivative applied on the host then the alpha=1 plan)."""
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    from dedalus_amd.tools import jacobi
    from oracle import np_transforms as npt
    rng = np.random.default_rng(78)
    shape, axis = {"contig": ((7, N), 1), "strided_even": ((3, N, 10), 1), "strided_odd": ((2, N, 7), 1)}[shape_kind]
    plan0, _ = _cheb_plan(N, M, 0)
    plan1, conv = _cheb_plan(N, M, 1)
    cs = list(shape)
    cs[axis] = M
    cin = rng.standard_normal(cs)
    D = jacobi.differentiation_matrix(M, -0.5, -0.5).toarray() * (2.0 / 1.7)
    assert np.count_nonzero(D - np.diag(np.diagonal(D, 1), 1)) == 0
    dvec = np.zeros(M)
    dvec[:M - 1] = np.diagonal(D, 1)
    dc = np.moveaxis(np.tensordot(D, np.moveaxis(cin, axis, 0), axes=(1, 0)), 0, axis)
    outer, inner = int(np.prod(shape[:axis])), int(np.prod(shape[axis + 1:]))
    d_c, d_dc, d_v = dev.from_host(cin), dev.from_host(np.ascontiguousarray(dc)), dev.from_host(dvec)
    outs = [dev.empty(shape, np.float64) for _ in range(4)]
    for o in outs:
        o.fill_(float("nan"))
    libhip.call("ddh_cheb_backward_dual", plan1, ptr(d_c), ptr(outs[0]), ptr(outs[1]), ptr(d_v), outer, inner, dev.stream)
    libhip.call("ddh_cheb_backward", plan0, ptr(d_c), ptr(outs[2]), outer, inner, dev.stream)
    libhip.call("ddh_cheb_backward", plan1, ptr(d_dc), ptr(outs[3]), outer, inner, dev.stream)
    dev.sync()
    g, gd, g1, gd1 = [dev.to_host(o) for o in outs]
    assert np.array_equal(g, g1)
    assert rel(gd, gd1) < 1e-14
    assert rel(g, npt.cheb_backward(cin, axis, N, None)) < TOL
    assert rel(gd, npt.cheb_backward(dc, axis, N, conv)) < TOL
    # a plan without conversion bands cannot serve the derivative pass
    with pytest.raises(Exception):
        libhip.call("ddh_cheb_backward_dual", plan0, ptr(d_c), ptr(outs[0]), ptr(outs[1]), ptr(d_v), outer, inner,
                    dev.stream)


@pytest.mark.parametrize("N,M", [(384, 256), (768, 512), (96, 64), (48, 64), (90, 60)])
@pytest.mark.parametrize("alpha", [0, 1, 2])
@pytest.mark.parametrize("shape_kind", ["contig", "strided_even", "strided_odd"])
def test_chebyshev_vs_oracle(dev, N, M, alpha, shape_kind):
    from oracle import np_transforms as npt
    rng = np.random.default_rng(4321)
    shape, axis = {"contig": ((5, N), 1), "strided_even": ((2, N, 12), 1), "strided_odd": ((2, N, 5), 1)}[shape_kind]
    plan, conv = _cheb_plan(N, M, alpha)
    g = rng.standard_normal(shape)
    cs = list(shape)
    cs[axis] = M
    assert rel(_run(dev, "ddh_cheb_forward", plan, g, cs, axis), npt.cheb_forward(g, axis, M, conv)) < TOL
    # smooth-ish coefficients so the ultraspherical back-substitution stays well conditioned
    cin = rng.standard_normal(cs) / (1.0 + np.arange(M).reshape([-1 if i == axis else 1 for i in range(len(cs))])) ** 2
    assert rel(_run(dev, "ddh_cheb_backward", plan, cin, shape, axis), npt.cheb_backward(cin, axis, N, conv)) < 1e-11


def test_round_trip_at_benchmark_line_sizes(dev):
    """Size-independent property at BASELINE sizes: forward(backward(c)) == c for resolved modes."""
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    t = dev.torch
    # z: [kz=256 -> 384] strided over 64x64 cells; y: contiguous 512 -> 768
    for (name, N, M, shape_c, axis) in [("cheb", 384, 256, (2, 256, 64 * 64), 1), ("rfft", 768, 512, (300, 512), 1),
                                        ("rfft", 768, 512, (3, 512, 130), 1)]:
        plan = _plan("ddh_plan_cheb", N, M, 0, None, None) if name == "cheb" else _plan("ddh_plan_rfft", N, M)
        gen = t.Generator(device=dev.tdev)
        gen.manual_seed(7)
        c = t.randn(shape_c, dtype=t.float64, device=dev.tdev, generator=gen)
        if name == "rfft":
            idx = [slice(None)] * len(shape_c)
            idx[axis] = 1
            c[tuple(idx)] = 0.0            # msin of k=0 is not a mode
        gs = list(shape_c)
        gs[axis] = N
        g = dev.empty(gs)
        c2 = dev.empty(shape_c)
        outer = int(np.prod(shape_c[:axis]))
        inner = int(np.prod(shape_c[axis + 1:]))
        libhip.call("ddh_%s_backward" % name, plan, ptr(c), ptr(g), outer, inner, dev.stream)
        libhip.call("ddh_%s_forward" % name, plan, ptr(g), ptr(c2), outer, inner, dev.stream)
        dev.sync()
        err = float((c2 - c).norm() / c.norm())
        assert err < TOL, (name, shape_c, err)


def test_mmt_matches_matrix_definition(dev):
    from dedalus_amd import libhip
    from oracle import np_transforms as npt
    rng = np.random.default_rng(5)
    N, M = 96, 64
    F, B = npt.chebyshev_mmt_matrices(N, M)
    h = C.c_uint64(0)
    Fc = np.ascontiguousarray(F)
    libhip.call("ddh_plan_mmt", C.byref(h), M, N, libhip.as_dp(Fc))
    g = rng.standard_normal((3, N, 37))
    out = _run(dev, "ddh_mmt_apply", h, g, (3, M, 37), 1)
    assert rel(out, npt.apply_matrix_along_axis(F, g, 1)) < TOL


@pytest.mark.parametrize("N,M,nlines", [(768, 512, 37), (24, 16, 10), (96, 64, 1), (1536, 1024, 6), (60, 40, 129),
                                        (384, 256, 64), (256, 170, 5), (512, 340, 9), (1024, 682, 3), (768, 640, 7),
                                        (384, 384, 2), (768, 40, 3)])
@pytest.mark.parametrize("na,nbs,ncs", [(3, (3, 9), (1, 3)), (1, (1,), (1,)), (2, (2, 4), (1, 2))])
def test_fused_grid_stage(dev, N, M, nlines, na, nbs, ncs):
    """ddh_rfft_bilinear_fused == backward transforms + products + forward transform of the oracle
    (u.grad(b) and u.grad(u) shaped term tables)."""
    from dedalus_amd.executor import HipExecutor
    from oracle.np_executor import NumpyExecutor
    rng = np.random.default_rng(N + 7 * nlines + na)
    a = rng.standard_normal((na, nlines, M))
    bs = [rng.standard_normal((nb, nlines, M)) for nb in nbs]
    for arr in [a] + bs:
        arr[..., 1] = 0.0                      # msin slot of k = 0 carries no data
    terms, ob, bb = [], 0, 0
    for nb, nc in zip(nbs, ncs):
        for c in range(nc):
            for j in range(na):
                terms.append((ob + c, j, bb + (c * na + j) % nb, float(rng.standard_normal())))
        ob += nc
        bb += nb
    spec = ("rfft", N, M)
    nex = NumpyExecutor()
    a_l = [a[i] for i in range(na)]
    b_l = [b[i] for b in bs for i in range(b.shape[0])]
    ref = [np.full((nlines, M), np.nan) for _ in range(ob)]
    ads = [0.0] * na
    bds = [(1.7 if i % 3 == 1 else 0.0) for i in range(len(b_l))]      # some operands differentiated at load
    if na > 1:
        ads[1] = 0.6
    nex.rfft_bilinear_fused(spec, None, a_l, b_l, ref, nlines, terms, ads, bds)
    hex_ = HipExecutor(dev)
    da = hex_.from_host(a)
    dbs = [hex_.from_host(b) for b in bs]
    dout = hex_.empty((ob, nlines, M))
    dout.fill_(float("nan"))
    hex_.rfft_bilinear_fused(spec, None, [da[i] for i in range(na)], [d[i] for d in dbs for i in range(d.shape[0])],
                             [dout[i] for i in range(ob)], nlines, terms, ads, bds)
    hex_.sync()
    got = hex_.download(dout)
    for i in range(ob):
        assert rel(got[i], ref[i]) < TOL, (i, rel(got[i], ref[i]))
