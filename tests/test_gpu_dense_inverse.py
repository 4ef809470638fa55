"""GPU: ddh_dense_inverse_* (device factorization of the sphere's per-m and the shell's per-ell subproblems) against
NumPy on random systems with valid-mode masks, real and complex, several sizes per batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cx", [False, True])
def test_batched_inverse_matches_numpy(cx):
    from dedalus_amd.executor import HipExecutor
    ex = HipExecutor()
    rng = np.random.default_rng(17 + cx)
    sizes = [1, 7, 64, 130, 301, 0, 33]
    Ms, Ls, rvs, cvs = [], [], [], []
    for n in sizes:
        def rnd():
            a = rng.standard_normal((n, n))
            return a + 1j * rng.standard_normal((n, n)) if cx else a
        M = rnd() + 3.0 * np.eye(n)
        L = rnd()
        rv = rng.random(n) > 0.15
        cv = np.zeros(n, dtype=bool)
        cv[rng.permutation(n)[:int(rv.sum())]] = True          # as many valid columns as valid rows, elsewhere
        Ms.append(M); Ls.append(L); rvs.append(rv); cvs.append(cv)
    inv = ex.make_dense_inverse(Ms, Ls, rvs, cvs, complex_=cx)
    for (a, b) in ((1.0, 0.37), (0.0, 1.0)):
        flat = ex.download(inv.compute(a, b))
        flat = flat.view(np.complex128) if cx else flat
        pos = 0
        for n, M, L, rv, cv in zip(sizes, Ms, Ls, rvs, cvs):
            got = flat[pos:pos + n * n].reshape(n, n)
            pos += n * n
            ref = np.zeros((n, n), dtype=got.dtype)
            if rv.any():
                ref[np.ix_(cv, rv)] = np.linalg.inv((a * M + b * L)[np.ix_(rv, cv)])
            assert np.linalg.norm(got - ref) <= 1e-11 * max(np.linalg.norm(ref), 1.0), (n, a, b)


@pytest.mark.parametrize("cx,n", [(False, 1289), (True, 1100), (False, 1025)])
def test_large_system_goes_through_the_multi_launch_path(cx, n):
    """n > 1024 (the mean-mode pencil of the 3-D Rayleigh-Benard problem is 1289 x 1289): panel + chip-wide rank-16 update
    launches instead of one workgroup, mixed with small systems in one batch, masks included."""
    from dedalus_amd.executor import HipExecutor
    ex = HipExecutor()
    rng = np.random.default_rng(n + cx)
    sizes = [40, n, 3]
    Ms, Ls, rvs, cvs = [], [], [], []
    for m in sizes:
        def rnd():
            a = rng.standard_normal((m, m))
            return a + 1j * rng.standard_normal((m, m)) if cx else a
        M = rnd() / np.sqrt(m) + 2.0 * np.eye(m)
        L = rnd() / np.sqrt(m)
        rv = rng.random(m) > 0.05
        cv = np.zeros(m, dtype=bool)
        cv[rng.permutation(m)[:int(rv.sum())]] = True
        Ms.append(M); Ls.append(L); rvs.append(rv); cvs.append(cv)
    inv = ex.make_dense_inverse(Ms, Ls, rvs, cvs, complex_=cx)
    for (a, b) in ((1.0, 0.37), (0.5, 1.0)):
        flat = ex.download(inv.compute(a, b))
        flat = flat.view(np.complex128) if cx else flat
        pos = 0
        for m, M, L, rv, cv in zip(sizes, Ms, Ls, rvs, cvs):
            got = flat[pos:pos + m * m].reshape(m, m)
            pos += m * m
            ref = np.zeros((m, m), dtype=got.dtype)
            ref[np.ix_(cv, rv)] = np.linalg.inv((a * M + b * L)[np.ix_(rv, cv)])
            assert np.linalg.norm(got - ref) <= 1e-10 * max(np.linalg.norm(ref), 1.0), (m, a, b)


def test_singular_system_is_reported():
    from dedalus_amd import libhip
    from dedalus_amd.executor import HipExecutor
    ex = HipExecutor()
    M = np.ones((4, 4))
    inv = ex.make_dense_inverse([M], [M], [np.ones(4, bool)], [np.ones(4, bool)])
    with pytest.raises(libhip.DdhError):
        inv.compute(1.0, 1.0)


@pytest.mark.parametrize("Nz", [24, 64])
def test_mean_mode_pencil_inverse_through_the_band_lu(Nz, monkeypatch):
    """The k = 0 pencil of 3-D Rayleigh-Benard (pressure gauge: the one pencil whose band block is singular) inverted by
    executor.BorderedBandInverse -- band LU + unit solves + ddh_ellband_bordered_inverse -- against numpy's inverse of
    the same matrix, and whole runs with timestep changes against the dense Gauss-Jordan path (DDH_FLAG_DENSE=1)."""
    import problems
    import dedalus_amd.public as d3
    from dedalus_amd.executor import BorderedBandInverse
    s, f = problems.rayleigh_benard_3d(d3, Nx=8, Ny=8, Nz=Nz)
    mats = [s.pack.matrices[mid].dense(0.0, 0.0, 0, 0, 1).real[np.ix_(s.row_perm, s.col_perm)] for mid in (s.MP_id, s.LP_id)]
    Md, Ld = mats
    n = s.n_interior
    zc = np.flatnonzero((np.abs(Md[:n, :n]) + np.abs(Ld[:n, :n])).sum(axis=0) == 0.0)
    assert len(zc) == 1
    bb = BorderedBandInverse(s.ex, Md, Ld, n, int(zc[0]))
    for (a, b) in ((1.0, 1e-3), (1.0, 0.3e-3)):
        got = s.ex.download(bb.compute(a, b)).reshape(n + 1, n + 1)
        ref = np.linalg.inv(a * Md + b * Ld)
        assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
    # the product takes this path by default: the cache entry of the factorization says so
    for dt in (1e-3, 1e-3, 2e-3, 1.5e-3):
        s.step(dt)
    assert any(v[0] == "band" for v in s.pack._flag_dinv.values())
    monkeypatch.setenv("DDH_FLAG_DENSE", "1")
    s2, f2 = problems.rayleigh_benard_3d(d3, Nx=8, Ny=8, Nz=Nz)
    for dt in (1e-3, 1e-3, 2e-3, 1.5e-3):
        s2.step(dt)
    assert all(v[0] == "dev" for v in s2.pack._flag_dinv.values())
    for k in ("p", "b", "u"):
        x, y = np.array(f[k]["c"]), np.array(f2[k]["c"])
        assert np.linalg.norm(x - y) <= 1e-10 * np.linalg.norm(y), k
