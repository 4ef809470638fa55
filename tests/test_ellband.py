"""Band structure of the shell's per-ell systems (dedalus_amd/core/ellband.py): the permutation + recombination found on
the host must reproduce the dense solutions when LAPACK's band LU stands in for the device kernels (CPU), and the device
band LU + sweeps (csrc/ddh_ellband.hip) must reproduce them on the GPU."""

import os

import numpy as np
import pytest

import problems


def _plan(s):
    from dedalus_amd.core.ellband import EllBandPlan
    prow = sorted({sc for m in s.emap for (sc, off, nr) in m if nr != s.Nr})
    pcol = sorted({sc for m in s.vmap for (sc, off, nr) in m if nr != s.Nr})
    return EllBandPlan(lambda g: s._dense(s.M_tl, g), lambda g: s._dense(s.L_tl, g),
                       [s.row_valid[:, g, :] for g in range(s.nl)], [s.col_valid[:, g, :] for g in range(s.nl)],
                       prow, pcol, s.Nr, range(s.nl))


def _smooth_solution(s, g, rng, ncol):
    cv = s.col_valid[:, g, :].reshape(-1)
    xt = np.zeros((s.R * s.Nr, ncol))
    xt[cv] = rng.standard_normal((cv.sum(), ncol)) * np.exp(-0.2 * (np.flatnonzero(cv) % s.Nr))[:, None]
    return xt


def test_band_plan_of_the_shell_convection_systems_reproduces_the_dense_solutions():
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    s, _ = problems.shell_convection(d3, shape=(16, 8, 32), dist_kw=dict(executor=NumpyExecutor()))
    plan = _plan(s)
    # every ell is a band of the same width (ell = 0 too: its gauge row is a boundary row of the pressure, and tau_p sits
    # next to the n = 0 row it enters)
    assert plan.dense_groups == []
    assert sorted(plan.per) == list(range(s.nl))
    assert plan.kl <= 24 and plan.kl + plan.ku <= 56 and plan.nbc == 8 and plan.mp <= 16
    assert plan.nmax < s.R * s.Nr
    a, b = 1.0, 0.05 * 2 / 3
    rng = np.random.default_rng(3)
    for g in range(s.nl):
        A = a * s._dense(s.M_tl, g) + b * s._dense(s.L_tl, g)
        rv = s.row_valid[:, g, :].reshape(-1)
        xt = _smooth_solution(s, g, rng, 3)
        rhs = A @ xt
        rhs[~rv] = 0.0
        x = plan.reference_solve(g, a, b, rhs)
        assert np.abs(x - xt).max() <= 2e-9 * np.abs(xt).max(), g
        # residual of the ORIGINAL system: the band path is a solve of the reference's matrix, not of a nearby one
        res = (A @ x - rhs)[rv]
        assert np.abs(res).max() <= 1e-13 * np.abs(rhs).max(), g


def test_band_plan_rejects_what_it_cannot_band():
    from dedalus_amd.core.ellband import EllBandPlan
    n = 8
    M = np.eye(2 * n)
    L = np.ones((2 * n, 2 * n))                     # dense coupling: no band
    plan = EllBandPlan(lambda g: M, lambda g: L, [np.ones((2, n), bool)], [np.ones((2, n), bool)], [], [], n, [0],
                       kl_max=4, w_max=8)
    assert plan.dense_groups == [0] and not plan.per


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 8, 32), (32, 16, 24)])
def test_device_band_lu_of_the_shell_systems_against_dense_solves(shape):
    import dedalus_amd.public as d3
    s, _ = problems.shell_convection(d3, shape=shape)
    assert s.ex.name == "hip"
    a, b = 1.0, 0.05 * 2 / 3
    lu = s.factor(a, b)
    assert isinstance(s._lus[lu], dict) and s._band["plan"].dense_groups == []         # the band path is the one that runs
    rng = np.random.default_rng(5)
    R, S, nl, Nr = s.R, 2 * s.nm, s.nl, s.Nr
    X = np.zeros((R, S, nl, Nr))
    RHS = np.zeros((R, S, nl, Nr))
    for g in range(nl):
        A = a * s._dense(s.M_tl, g) + b * s._dense(s.L_tl, g)
        rv = s.row_valid[:, g, :].reshape(-1)
        nslot = 2 * min(g + 1, s.nm)
        xt = _smooth_solution(s, g, rng, nslot)
        rhs = A @ xt
        rhs[~rv] = 0.0
        X[:, :nslot, g, :] = xt.reshape(R, Nr, nslot).transpose(0, 2, 1)
        RHS[:, :nslot, g, :] = rhs.reshape(R, Nr, nslot).transpose(0, 2, 1)
    rhs_d = s.ex.from_host(RHS.reshape(R, S, nl * Nr))
    x_d = s.ex.zeros((R, S, nl * Nr))
    s.ex.assign(x_d, s.ex.from_host(np.full((R, S, nl * Nr), 7.0)))          # stale content must not survive
    s.solve(lu, rhs_d, x_d)
    x = np.asarray(s.ex.download(x_d)).reshape(R, S, nl, Nr)
    for g in range(nl):
        err = np.abs(x[:, :, g] - X[:, :, g]).max() / np.abs(X[:, :, g]).max()
        assert err <= 5e-9, (g, err)
    # a second factorization (another timestep) refills the same storage and gives the other solution
    lu2 = s.factor(a, 0.5 * b, reuse=lu)
    assert lu2 == lu
    s.solve(lu, rhs_d, x_d)
    x2 = np.asarray(s.ex.download(x_d)).reshape(R, S, nl, Nr)
    g = nl - 1
    A2 = a * s._dense(s.M_tl, g) + 0.5 * b * s._dense(s.L_tl, g)
    rv, cv = s.row_valid[:, g, :].reshape(-1), s.col_valid[:, g, :].reshape(-1)
    ref = np.zeros((R * Nr, S))
    ref[cv] = np.linalg.solve(A2[np.ix_(rv, cv)], RHS[:, :, g, :].transpose(0, 2, 1).reshape(R * Nr, S)[rv])
    got = x2[:, :, g, :].transpose(0, 2, 1).reshape(R * Nr, S)
    assert np.abs(got - ref).max() <= 5e-9 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("ts", ["SBDF2", "RK443"])
def test_band_and_dense_paths_take_the_same_steps(monkeypatch, ts):
    """multistep (one factorization, refilled when the timestep changes) and Runge-Kutta (one factorization per distinct
    diagonal entry of H, all alive at once)"""
    import dedalus_amd.public as d3
    states = []
    for dense in ("0", "1"):
        monkeypatch.setenv("DDH_SHELL_DENSE", dense)
        s, fields = problems.shell_convection(d3, shape=(16, 8, 16), timestepper=ts)
        for k in range(5):
            s.step(0.02 if k < 3 else 0.015)
        assert bool(s._band) == (dense == "0")
        states.append({f.name: np.array(f["c"]) for f in s.state if hasattr(f, "basis")})
    worst = {}
    for name in states[0]:
        x, y = states[0][name], states[1][name]
        worst[name] = np.abs(x - y).max() / max(np.abs(y).max(), 1e-8)
    # the pressure and the tau fields are the ill-conditioned unknowns (cond ~ 1e8 per system: they move in the 8th digit
    # between two summation orders of the SAME dense inverse); buoyancy and velocity agree to rounding
    assert worst["b"] <= 1e-10 and worst["u"] <= 1e-10, worst
    assert max(worst.values()) <= 1e-6, worst


def test_sphere_inverses_from_the_band_plan_of_the_real_form_transposed_systems():
    """(a M + b L)_m^-1 row by row from unit solves of the real-form TRANSPOSED band systems (LAPACK standing in for the
    device kernels) against numpy's inverse of the complex matrices"""
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    from dedalus_amd.core.ellband import EllBandPlan
    s, _, _ = problems.shallow_water(d3, Nphi=32, Ntheta=16, dist_kw=dict(executor=NumpyExecutor()))
    R, nl, nm = s.R, s.basis.nl, s.basis.nm
    above = lambda m: (np.arange(nl) >= m)[None, :]
    rv = [np.repeat(s.row_valid[:, m, :] * above(m), 2, axis=0) for m in range(nm)]
    cv = [np.repeat(s.col_valid[:, m, :] * above(m), 2, axis=0) for m in range(nm)]
    plan = EllBandPlan(lambda m: s._real_form_transposed(s.M_tl, m), lambda m: s._real_form_transposed(s.L_tl, m),
                       cv, rv, [], [], nl, range(nm))
    assert not plan.dense_groups and plan.nbc == 0 and plan.mp == 0 and plan.kl <= 11 and plan.ku <= 11
    a, b = 1.0, 0.37
    for m in (0, 1, nm // 2, nm - 2):
        ne = nl - m
        if ne <= 0 or plan.n[m] == 0:
            continue
        A = a * s._dense(s.M_tl, m) + b * s._dense(s.L_tl, m)
        rvm, cvm = s.row_valid[:, m, m:].reshape(-1), s.col_valid[:, m, m:].reshape(-1)
        B = np.zeros_like(A)
        B[np.ix_(cvm, rvm)] = np.linalg.inv(A[np.ix_(rvm, cvm)])
        rhs = np.zeros((2 * R * nl, R * ne))
        for c in range(R):
            for e in range(ne):
                rhs[(2 * c) * nl + m + e, c * ne + e] = 1.0
        X = plan.reference_solve(m, a, b, rhs)                   # column j: A^-T e_j = row j of the inverse
        got = np.zeros_like(B)
        for cp in range(R):
            got[:, cp * ne:(cp + 1) * ne] = (X[(2 * cp) * nl + m:(2 * cp) * nl + nl] + 1j * X[(2 * cp + 1) * nl + m:(2 * cp + 1) * nl + nl]).T
        assert np.abs(got - B).max() <= 1e-13 * np.abs(B).max(), m


@pytest.mark.gpu
def test_sphere_band_and_dense_inverses_take_the_same_steps(monkeypatch):
    import dedalus_amd.public as d3
    states = []
    for dense in ("0", "1"):
        monkeypatch.setenv("DDH_SPHERE_DENSE", dense)
        s, fields, extra = problems.shallow_water(d3, Nphi=64, Ntheta=32)
        dt = extra["timestep"]
        for k in range(6):
            s.step(dt if k < 3 else 0.7 * dt)                   # (a timestep change: the inverses are formed again)
        assert bool(s._sband) == (dense == "0")
        states.append({f.name: np.array(f["c"]) for f in s.state})
    for name in states[0]:
        x, y = states[0][name], states[1][name]
        assert np.abs(x - y).max() <= 1e-12 * max(np.abs(y).max(), 1e-12), name
