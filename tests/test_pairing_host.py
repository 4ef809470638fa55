"""Host logic of the partner pencils (dedalus_amd/core/solvers.py _build_pairing, ddh_pencil_set_pairing): the x <-> y symmetry
is detected on the term lists, and the relations the device kernels rely on hold for the assembled pencil matrices of the
CPU oracle:   lambda(ky, kx) = Pi_r lambda(kx, ky) Pi_c   (the +kx systems),
              lambda(-ky, kx) = Pi_r conj(lambda(-kx, ky)) Pi_c   (the -kx systems: conjugated data in and out)."""
import numpy as np
import pytest

import problems


def _solver(**kw):
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    return problems.rayleigh_benard_3d(d3, dist_kw=dict(executor=NumpyExecutor()), **kw)[0]


def test_symmetric_box_is_detected_and_the_matrix_relations_hold():
    s = _solver(Nx=8, Ny=8, Nz=8)
    assert s.pairing is not None
    rs, cs = s.pairing["row_swap"], s.pairing["col_swap"]
    assert np.array_equal(rs[rs], np.arange(s.R)) and np.array_equal(cs[cs], np.arange(s.R))
    assert not np.array_equal(cs, np.arange(s.R))              # the velocity components do swap
    assert s.pack.pairing is not None                          # handed to the pack
    kx, ky = s.pack.kx, s.pack.ky
    without_conj = 0.0
    for mid in (s.MP_id, s.LP_id):
        A = s.pack.mats[mid]
        for mx, my in [(1, 2), (3, 1), (2, 3)]:
            A1 = A.matrix(kx[mx], ky[my], mx, my, 1).toarray()
            A2 = A.matrix(kx[my], ky[mx], my, mx, 1).toarray()
            assert np.abs(A2 - A1[rs][:, cs]).max() <= 1e-13 * np.abs(A1).max()
            Q1 = A.matrix(kx[mx], ky[my], mx, my, -1).toarray()
            Q2 = A.matrix(kx[my], ky[mx], my, mx, -1).toarray()
            assert np.abs(Q2 - np.conj(Q1)[rs][:, cs]).max() <= 1e-13 * np.abs(Q1).max()
            without_conj = max(without_conj, np.abs(Q2 - Q1[rs][:, cs]).max())
    assert without_conj > 1e-3                                  # (without the conjugation the -kx relation does not hold)


@pytest.mark.parametrize("kw", [dict(Nx=8, Ny=8, Nz=8, Lx=4, Ly=2), dict(Nx=8, Ny=12, Nz=8)])
def test_anisotropic_boxes_are_left_alone(kw):
    assert _solver(**kw).pairing is None


def test_two_dimensional_problems_are_left_alone():
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    s, f = problems.rayleigh_benard_2d(d3, Nx=16, Nz=8, dist_kw=dict(executor=NumpyExecutor()))
    assert s.pairing is None


def test_switch(monkeypatch):
    monkeypatch.setenv("DDH_PAIR", "0")
    assert _solver(Nx=8, Ny=8, Nz=8).pairing is None


def test_bordered_band_inverse_of_the_mean_mode_pencil():
    """Host restatement of ddh_ellband_bordered_inverse (csrc/ddh_ellband.hip) on the solver's own k = 0 pencil of 3-D
    Rayleigh-Benard: the band block is singular only through the vanishing column of the constant pressure mode; with
    that column exchanged for tau_p's the matrix is block triangular around a band, and the inverse assembled from the
    band block's inverse equals the dense inverse."""
    import problems
    import dedalus_amd.public as d3
    from dedalus_amd.pencilpack import TermList
    from dedalus_amd.core.ellband import BandBlockPlan
    from oracle.np_executor import NumpyExecutor
    s, _ = problems.rayleigh_benard_3d(d3, Nx=8, Ny=8, Nz=24, dist_kw=dict(executor=NumpyExecutor()))
    mats = []
    for mid in (s.MP_id, s.LP_id):
        t = s.pack.mats[mid]
        A = TermList(t.nrows, t.ncols, t.row, t.col, t.coef, t.ex, t.ey, t.dx, t.dy).dense(0.0, 0.0, 0, 0, 1)
        assert np.abs(A.imag).max() == 0.0
        mats.append(A.real[np.ix_(s.row_perm, s.col_perm)])
    Md, Ld = mats
    n, N = s.n_interior, s.R
    assert N - n == 1
    zc = np.flatnonzero((np.abs(Md[:n, :n]) + np.abs(Ld[:n, :n])).sum(axis=0) == 0.0)
    assert len(zc) == 1
    j0 = int(zc[0])
    M2, L2 = Md[:n, :n].copy(), Ld[:n, :n].copy()
    M2[:, j0], L2[:, j0] = Md[:n, n], Ld[:n, n]
    plan = BandBlockPlan(M2, L2)
    assert plan.kl <= s.kl + 2 and plan.ku <= s.ku + 2          # still the solver's narrow band
    for (a, b) in ((1.0, 1e-3), (1.0, 0.25e-3), (1.5, 2e-3)):
        A = a * Md + b * Ld
        X = np.linalg.inv(a * M2 + b * L2)
        w = A[n, :n].copy()
        d = w[j0]
        w[j0] = A[n, n]
        inv = np.zeros((N, N))
        inv[:n, :n] = X
        inv[n, :n] = X[j0]
        inv[j0, :n] = -(w @ X) / d
        inv[j0, n] = 1.0 / d
        ref = np.linalg.inv(A)
        assert np.abs(inv - ref).max() <= 1e-9 * np.abs(ref).max()


def test_block_inverse_plan_holds_the_transposed_diagonal_blocks():
    """Host analysis of the few-system path (SolverBase._block_inverse_plan): for 2-D Rayleigh-Benard the band arrays
    handed to the numeric band LU are, per (pencil, parity block), the TRANSPOSE of that diagonal block of the graded real
    matrix a M + b L in the solver's permuted order -- so that unit solve s yields row s of the block's inverse."""
    import problems
    import dedalus_amd.public as d3
    from dedalus_amd.pencilpack import TermList
    from oracle.np_executor import NumpyExecutor
    s, _ = problems.rayleigh_benard_2d(d3, Nx=16, Nz=64, dist_kw=dict(executor=NumpyExecutor()))
    # (the oracle's pack keeps term lists under another name)
    s.pack.matrices = [TermList(t.nrows, t.ncols, t.row, t.col, t.coef, t.ex, t.ey, t.dx, t.dy) for t in s.pack.mats]
    assert s.allow_block_inverse
    bi = s._block_inverse_plan(any_executor=True)
    assert bi, "2-D Rayleigh-Benard must be eligible"
    plan, ns, nh = bi["plan"], bi["ns"], bi["nh"]
    assert ns == 2 and ns * nh == s.n_interior and plan.nl == (s.nx // 2) * ns
    rg = s.real_grading
    a, b = 1.0, 0.5e-3
    for cell in (1, 3, 7):
        kx = s.pack.kx[cell]
        A = (a * s.pack.matrices[rg["matM"]].dense(kx, 0.0, cell, 0, 1) + b * s.pack.matrices[rg["matL"]].dense(kx, 0.0, cell, 0, 1))
        assert np.abs(A.imag).max() == 0.0
        A = A.real[np.ix_(s.row_perm, s.col_perm)][:s.n_interior, :s.n_interior]
        for blk in range(ns):
            Bt = A[blk * nh:(blk + 1) * nh, blk * nh:(blk + 1) * nh].T
            band = a * plan.MB[cell * ns + blk] + b * plan.LB[cell * ns + blk]
            dense = np.zeros((nh, nh))
            for i in range(nh):
                for d in range(plan.kl + plan.ku + 1):
                    j = i - plan.kl + d
                    if 0 <= j < nh:
                        dense[i, j] = band[i, d]
            assert np.abs(dense - Bt).max() <= 1e-14 * np.abs(Bt).max()
        # nothing couples the two blocks
        assert np.abs(A[:nh, nh:]).max() == 0.0 and np.abs(A[nh:, :nh]).max() == 0.0
