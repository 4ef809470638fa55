"""CPU: the row masks of the Runge-Kutta solve (rows that are zero in every right-hand-side term; unknowns nothing reads
between stages) are what the problem structure says -- checked against the assembled vectors of the oracle executor."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import problems  # noqa: E402


def _solver():
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    return problems.rayleigh_benard_3d(d3, Nx=8, Ny=8, Nz=12, timestepper="RK222", dist_kw=dict(executor=NumpyExecutor()))


def test_zero_rows_are_zero_in_mx_and_f():
    solver, f = _solver()
    z = solver.zero_rows_host()
    if z is None:                                   # the oracle executor may take the general F path: nothing to check
        assert solver.F_direct is None
        return
    ex = solver.ex
    solver.sync_state_to_device()
    F = ex.zeros((solver.R, solver.nx, solver.ny))
    solver.evaluate_F(F)
    MX = ex.zeros((solver.R, solver.nx, solver.ny))
    solver.pack.matvec(solver.M_id, solver.X, MX)
    Fh, MXh = np.asarray(ex.download(F)), np.asarray(ex.download(MX))
    rows = np.flatnonzero(z)
    assert rows.size and np.all(Fh[rows] == 0.0) and np.all(MXh[rows] == 0.0)
    cont = solver.eq_info[0]
    assert z[cont["row0"]:cont["row0"] + cont["rows"]].all()
    # every row with data is unmasked
    busy = np.flatnonzero(np.abs(Fh).reshape(solver.R, -1).max(axis=1) + np.abs(MXh).reshape(solver.R, -1).max(axis=1))
    assert not z[busy].any()


def test_skip_rows_are_pressure_and_taus():
    solver, f = _solver()
    k = solver.skip_rows_host()
    for info in solver.var_info:
        want = info["field"].name not in ("b", "u")
        blk = k[info["row0"]:info["row0"] + info["rows"]]
        assert blk.all() == want and blk.any() == want, info["field"].name
