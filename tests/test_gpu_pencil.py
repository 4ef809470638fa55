"""GPU parity of the batched pencil engine (mat-vec, bordered band LU, solve, dense fallback)
against oracle/np_pencil.py (per-pencil scipy CSR + SuperLU, the reference's algorithm)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def build_problem(rng, nf, nvar, Nz, nbord, ncx, ncy, gauge=False):
    """Random tau-like system: nvar interior variables x Nz modes (upper-banded in the mode index,
    dense coupling between variables of the same mode), nbord border columns (lift-like, touching
    the last modes) and nbord dense border rows.  With gauge=True variable 0 is only defined up to
    a constant at k=0, fixed by an extra border row/column that exists only in the k=0 pencil."""
    n = nvar * Nz
    nb = nbord + (1 if gauge else 0)
    N = n + nb
    rows, cols, coef, ex, ey, dx, dy = [], [], [], [], [], [], []

    def add(r, c, v, e=(0, 0), d=(0, 0)):
        rows.append(r); cols.append(c); coef.append(v); ex.append(e[0]); ey.append(e[1]); dx.append(d[0]); dy.append(d[1])

    M = ([], [], [], [], [], [], [])
    Lm = (rows, cols, coef, ex, ey, dx, dy)
    phys = lambda v, kz: v * Nz + kz
    # mass matrix: conversion-like bands 0, +2 on variables 1..nvar-1 (variable 0 is "pressure")
    mr, mc, mv = [], [], []
    for v in range(1, nvar):
        for kz in range(Nz):
            mr.append(phys(v, kz)); mc.append(phys(v, kz)); mv.append(1.0 + 0.1 * rng.random())
            if kz + 2 < Nz:
                mr.append(phys(v, kz)); mc.append(phys(v, kz + 2)); mv.append(-0.4 * rng.random())
    # stiffness: k-dependent, upper banded in kz (offsets 0..3), couples all variables
    for ve in range(nvar):
        for kz in range(Nz):
            for vv in range(nvar):
                for off in range(0, 4):
                    if kz + off >= Nz:
                        continue
                    r, c = phys(ve, kz), phys(vv, kz + off)
                    if ve == vv and off == 0:
                        if ve == 0:
                            if nf >= 1:
                                add(r, c, 1.0 + rng.random(), (2, 0))
                            if nf == 2:
                                add(r, c, 1.0 + rng.random(), (0, 2))
                            if nf == 0 or not gauge:
                                add(r, c, 3.0)
                        else:
                            add(r, c, 6.0 + rng.random())
                            if nf >= 1:
                                add(r, c, 0.5, (2, 0))
                    else:
                        if rng.random() < 0.5:
                            add(r, c, 0.3 * (rng.random() - 0.5))
                        if nf >= 1 and rng.random() < 0.3:
                            add(r, c, 0.2j * (rng.random() - 0.5), (1, 0))
                        if nf == 2 and rng.random() < 0.3:
                            add(r, c, 0.2j * (rng.random() - 0.5), (0, 1))
    # border columns (taus): entries in the last two modes of one equation each
    for t in range(nbord):
        ve = t % nvar
        sh = t // nvar
        add(phys(ve, Nz - 1 - sh), n + t, 1.0)
        add(phys(ve, Nz - 3 - sh), n + t, -0.5)
    # border rows (boundary conditions): dense over the modes of one variable
    for t in range(nbord):
        vv = t % nvar
        for kz in range(Nz):
            add(n + t, phys(vv, kz), ((-1.0) ** (kz * (t // nvar)) if t >= nvar else 1.0) * (1.0 + 0.01 * kz))
    row_axes = np.full(N, 3, dtype=np.uint8)
    col_axes = np.full(N, 3, dtype=np.uint8)
    if gauge:
        # variable 0 at k=0 has a zero diagonal -> gauge row "sum of var 0 modes = 0" and a gauge column
        # in equation 0, both existing only for the k=0 pencil
        for kz in range(Nz):
            add(N - 1, phys(0, kz), 1.0 / (1 + kz), (0, 0), (1 if nf >= 1 else 0, 1 if nf == 2 else 0))
        add(phys(0, 0), N - 1, 1.0, (0, 0), (1 if nf >= 1 else 0, 1 if nf == 2 else 0))
        row_axes[N - 1] = 0
        col_axes[N - 1] = 0
    # logical ordering: mode-major interior, then border
    perm = np.array([phys(v, kz) for kz in range(Nz) for v in range(nvar)] + list(range(n, N)), dtype=np.int32)
    Mterms = (mr, mc, mv)
    return dict(n=n, nb=nb, N=N, M=Mterms, L=Lm, perm=perm, row_axes=row_axes, col_axes=col_axes)


def bandwidth(tlists, perm, n):
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    kl = ku = 0
    for tl in tlists:
        i, c = inv[tl.row], inv[tl.col]
        m = (i < n) & (tl.dx == 0) & (tl.dy == 0)
        if m.any():
            kl = max(kl, int((i[m] - c[m]).max()))
            ku = max(ku, int((c[m] - i[m]).max()))
    return kl, ku


@pytest.mark.parametrize("nf,ncx,ncy", [(2, 6, 10), (1, 40, 1), (0, 1, 1)])
@pytest.mark.parametrize("gauge", [False, True])
@pytest.mark.parametrize("coop", [0, 2, 4])
def test_matvec_factor_solve_vs_oracle(nf, ncx, ncy, gauge, coop):
    # coop = 2: the cooperative sweeps (16 lanes per system, used for few systems); 0: one thread per system;
    # 4: one thread per system forward, 4 lanes per system backward (the mid-range choice)
    from dedalus_amd.device import Device
    from dedalus_amd.pencilpack import PencilPack, TermList
    from oracle import np_pencil as npp
    dev = Device.get()
    rng = np.random.default_rng(99 + nf)
    nvar, Nz, nbord = 3, 12, 4
    pb = build_problem(rng, nf, nvar, Nz, nbord, ncx, ncy, gauge)
    N, n = pb["N"], pb["n"]
    nx = 2 * ncx if nf >= 1 else 1
    ny = 2 * ncy if nf == 2 else 1
    kx = 0.7 * np.arange(ncx)
    ky = 1.3 * np.arange(ncy)
    mr, mc, mv = pb["M"]
    M = TermList(N, N, mr, mc, mv)
    L = TermList(N, N, *pb["L"])
    Mo = npp.TermList(N, N, M.row, M.col, M.coef, M.ex, M.ey, M.dx, M.dy)
    Lo = npp.TermList(N, N, L.row, L.col, L.coef, L.ex, L.ey, L.dx, L.dy)
    pack = PencilPack(dev, nf, N, nx, ny, kx, ky)
    pack.set_solve_variant(0 if coop == 4 else coop, -1, 4 if coop == 4 else -1)
    idM, idL = pack.add_matrix(M), pack.add_matrix(L)
    # random data obeying the real-Fourier structure (msin parts of m=0 vanish)
    x = rng.standard_normal((N, nx, ny))
    if nf >= 1:
        x[:, 1, :] = 0.0
    if nf == 2:
        x[:, :, 1] = 0.0
    # rows/cols that exist only at k=0 carry data only there
    for r in range(N):
        if pb["col_axes"][r] == 0:
            keep = x[r, 0, 0]
            x[r] = 0.0
            x[r, 0, 0] = keep
    dx = dev.from_host(x)
    dy = dev.empty((N, nx, ny))
    pack.matvec(idL, dx, dy)
    dev.sync()
    y_ref = npp.matvec(Lo, x, nf, kx, ky)
    assert rel(dev.to_host(dy), y_ref) < 1e-13
    # factor + solve
    a, b = 1.0, 0.37
    kl, ku = bandwidth([M, L], pb["perm"], n) if nf > 0 else (0, pb["nb"])   # single pencil -> dense path
    lu = pack.factor(idM, idL, a, b, pb["perm"], pb["perm"], n, kl, ku, pb["row_axes"], pb["col_axes"])
    if gauge and nf > 0:
        assert pack.lu_meta[lu]["nflag"] >= 1      # the k=0 pencil has a singular band block
    rhs = rng.standard_normal((N, nx, ny))
    if nf >= 1:
        rhs[:, 1, :] = 0.0
    if nf == 2:
        rhs[:, :, 1] = 0.0
    for r in range(N):
        if pb["row_axes"][r] == 0:
            keep = rhs[r, 0, 0]
            rhs[r] = 0.0
            rhs[r, 0, 0] = keep
    drhs = dev.from_host(rhs)
    dsol = dev.empty((N, nx, ny))
    pack.solve(lu, drhs, dsol)
    dev.sync()
    ref = npp.PencilLU(Mo, Lo, a, b, nf, nx, ny, kx, ky, pb["row_axes"], pb["col_axes"]).solve(rhs)
    sol = dev.to_host(dsol)
    assert np.isfinite(sol).all()
    assert rel(sol, ref.reshape(sol.shape)) < 1e-11
    # residual check through the device mat-vec: (aM + bL) x == rhs
    d1, d2 = dev.empty((N, nx, ny)), dev.empty((N, nx, ny))
    pack.matvec(idM, dsol, d1)
    pack.matvec(idL, dsol, d2)
    dev.sync()
    res = a * dev.to_host(d1) + b * dev.to_host(d2)
    # rows that do not exist for a pencil are identity rows there: mask them out of the residual
    mask = np.ones_like(rhs, dtype=bool)
    for r in range(N):
        if pb["row_axes"][r] == 0:
            mask[r] = False
            mask[r, 0, 0] = True
    assert rel(res[mask], rhs[mask]) < 1e-10
    # re-factor in place with a different coefficient (dt change) and solve again
    lu2 = pack.factor(idM, idL, a, 0.11, pb["perm"], pb["perm"], n, kl, ku, pb["row_axes"], pb["col_axes"], reuse=lu)
    assert lu2 == lu
    pack.solve(lu, drhs, dsol)
    dev.sync()
    ref2 = npp.PencilLU(Mo, Lo, a, 0.11, nf, nx, ny, kx, ky, pb["row_axes"], pb["col_axes"]).solve(rhs)
    assert rel(dev.to_host(dsol), ref2.reshape(sol.shape)) < 1e-11


def test_lincomb_bilinear_cfl_a2a():
    import ctypes as C
    from dedalus_amd import libhip
    from dedalus_amd.device import Device, ptr
    dev = Device.get()
    rng = np.random.default_rng(3)
    n = 100003
    xs = [rng.standard_normal(n) for _ in range(5)]
    al = np.array([1.0, -0.5, 2.0, 0.25, -3.0])
    dxs = [dev.from_host(x) for x in xs]
    dy = dev.empty(n)
    arr = (C.c_void_p * 5)(*[C.c_void_p(d.data_ptr()) for d in dxs])
    libhip.call("ddh_lincomb", ptr(dy), 5, arr, libhip.as_dp(al), n, dev.stream)
    dev.sync()
    assert rel(dev.to_host(dy), sum(a * x for a, x in zip(al, xs))) < 1e-14
    # dot product u.grad(b) with 3 components
    npts = 4096
    u = rng.standard_normal((3, npts))
    g = rng.standard_normal((3, npts))
    out = dev.empty((1, npts))
    ic = np.zeros(3, np.int32); ia = np.arange(3, dtype=np.int32); ib = np.arange(3, dtype=np.int32)
    cf = -np.ones(3)
    d_u, d_g = dev.from_host(u), dev.from_host(g)
    libhip.call("ddh_grid_bilinear", ptr(out), 1, ptr(d_u), ptr(d_g), npts, 3,
                libhip.as_ip(ic), libhip.as_ip(ia), libhip.as_ip(ib), libhip.as_dp(cf), dev.stream)
    dev.sync()
    assert rel(dev.to_host(out)[0], -(u * g).sum(0)) < 1e-14
    # a2a pack / unpack are inverse re-orderings of a transpose (even and odd segment lengths)
    for outer, na, nb_, inner, P in [(2, 8, 12, 6, 4), (3, 6, 1, 1, 2), (2, 10, 3, 5, 1), (5, 4, 6, 1, 2)]:
        src = rng.standard_normal((outer, na, nb_, inner))
        d_src = dev.from_host(src)
        d_pk = dev.empty(src.size)
        libhip.call("ddh_a2a_pack", ptr(d_src), ptr(d_pk), outer, na, nb_, inner, P, dev.stream)
        dev.sync()
        pk = dev.to_host(d_pk).reshape(P, outer, na // P, nb_, inner)
        for p in range(P):
            assert np.array_equal(pk[p], src[:, p * (na // P):(p + 1) * (na // P)])
        if nb_ % P:
            continue
        blocks = rng.standard_normal((P, outer, na, nb_ // P, inner))
        d_un = dev.empty((outer, na, nb_, inner))
        d_blocks = dev.from_host(blocks)
        libhip.call("ddh_a2a_unpack", ptr(d_blocks), ptr(d_un), outer, na, nb_, inner, P, dev.stream)
        dev.sync()
        un = dev.to_host(d_un)
        for p in range(P):
            assert np.array_equal(un[:, :, p * (nb_ // P):(p + 1) * (nb_ // P)], blocks[p])


def test_window_matvec_equals_the_general_kernel_bit_for_bit(monkeypatch):
    """A real, wavenumber-independent banded matrix (the mass matrix of an IVP) takes band_matvec_kernel from 16 384 cells
    on: same result as the term-list kernel bit for bit, and equal to the dense product applied to every mode."""
    from dedalus_amd.device import Device
    from dedalus_amd.pencilpack import PencilPack, TermList
    dev = Device.get()
    rng = np.random.default_rng(5)
    nvar, Nz, ncx, ncy = 3, 40, 128, 128
    N = nvar * Nz + 5
    rows, cols, vals = [], [], []
    for v in range(1, nvar):                       # variable 0 and the 5 border rows have no terms (zero rows of y)
        for kz in range(Nz):
            for off in (0, 2, 4):
                if kz + off < Nz and (off == 0 or rng.random() < 0.8):
                    rows.append(v * Nz + kz); cols.append(v * Nz + kz + off); vals.append(rng.standard_normal())
    M = TermList(N, N, rows, cols, vals)
    nx, ny = 2 * ncx, 2 * ncy
    pack = PencilPack(dev, 2, N, nx, ny, 0.7 * np.arange(ncx), 1.3 * np.arange(ncy))
    id_band = pack.add_matrix(M)
    monkeypatch.setenv("DDH_MV_NOBAND", "1")
    id_plain = pack.add_matrix(M)
    monkeypatch.delenv("DDH_MV_NOBAND")
    x = rng.standard_normal((N, nx, ny))
    xd = dev.from_host(x)
    y1, y2 = dev.empty((N, nx, ny)), dev.empty((N, nx, ny))
    y1.fill_(7.0); y2.fill_(-3.0)
    pack.matvec(id_band, xd, y1)
    pack.matvec(id_plain, xd, y2)
    a, b = y1.cpu().numpy(), y2.cpu().numpy()
    assert np.array_equal(a, b)
    dense = np.zeros((N, N))
    np.add.at(dense, (np.array(rows), np.array(cols)), np.array(vals))
    want = np.einsum("rc,cxy->rxy", dense, x)
    assert rel(a, want) < 1e-14
    assert np.all(a[:Nz] == 0.0) and np.all(a[nvar * Nz:] == 0.0)
