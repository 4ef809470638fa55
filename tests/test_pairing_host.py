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
