"""The explicit half of a 3-D Rayleigh-Benard stage against the reference at ANY resolution (test infrastructure).

tests/golden/config_explicit.npz holds F = (0, -u.grad(b), -u.grad(u), boundary constants) of the UNMODIFIED reference
(oracle/make_golden_config.py::config_explicit) for a band-limited state at 16^3 modes.  The same state at any other
resolution has the same coefficients in the modes both carry and zeros elsewhere (the products are exact under 3/2
dealiasing, the conversions to the equation bases are polynomial identities), so one small golden pins the whole chain
-- backward z / x transforms, fused y stage with derivatives at load, forward transforms writing the equation rows of
the F system vector -- at the metric's 512 x 512 x 256 too."""
import os

import numpy as np

from oracle.make_golden_config import band_limited_state

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_explicit.npz")


def check(solver, fields, tol=1e-13):
    gold = np.load(GOLD)
    Ns = int(gold["N"])
    grids = solver.dist.local_grids(*fields["b"].domain.bases)
    band_limited_state(fields, grids)
    solver.sync_state_to_device()
    out = solver.ex.zeros((solver.R, solver.nx, solver.ny))
    solver.evaluate_F(out)
    F = solver.equation_space_to_user(out)       # (boundary rows in the user's equations, SolverBase.eq_T)
    scale = max(float(np.abs(gold["F1"]).max()), float(np.abs(gold["F2"]).max()))
    worst = 0.0
    for i, info in enumerate(solver.eq_info):
        ref = gold["F%d" % i]
        nz, nc = info["nz"], info["ncomp"]
        blk = F[info["row0"]:info["row0"] + info["rows"]].reshape(nc, nz, solver.nx, solver.ny)
        ref = ref.reshape((nc,) + ref.shape[-3:])                     # [comp][x][y][z]
        sx, sy, sz = min(ref.shape[1], solver.nx), min(ref.shape[2], solver.ny), min(ref.shape[3], nz)
        # everything the golden holds beyond the modes this solver carries must be (numerically) empty
        rest_ref = ref.copy()
        rest_ref[:, :sx, :sy, :sz] = 0.0
        assert np.abs(rest_ref).max() < 1e-14, ("golden populated beyond the solver's modes", i)
        want = np.zeros_like(blk)
        want[:, :sz, :sx, :sy] = np.transpose(ref[:, :sx, :sy, :sz], (0, 3, 1, 2))
        scl = scale if i in (1, 2) else 1.0
        err = float(np.abs(blk - want).max()) / scl
        worst = max(worst, err)
        assert err < tol, ("equation %d" % i, err, Ns)
    return worst
