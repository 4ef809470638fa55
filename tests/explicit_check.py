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


SHELL_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_shell_explicit.npz")


def shell_labels(F):
    """(m, ell, n) of every element of a shell field's coefficient array in this package's user layout (the reference's
    packed layout: SphereBasis.packed_groups restates core/basis.py:2868-2889)."""
    m, ell = F.basis.sphere.packed_groups()
    Nr = np.asarray(F["c"]).shape[-1]
    shp = m.shape + (Nr,)
    return (np.broadcast_to(m[:, :, None], shp), np.broadcast_to(ell[:, :, None], shp),
            np.broadcast_to(np.arange(Nr)[None, None, :], shp))


def check_shell(d3, shape, dist_kw=None, tol=1e-12):
    """The explicit half of a shell-convection step at `shape` against the reference's table for the band-limited state
    (oracle/make_golden_config.py::config_shell_explicit): every listed mode within tol of the largest coefficient of its
    equation, every unlisted mode below tol."""
    import problems
    gold = np.load(SHELL_GOLD)
    ref = {tuple(int(v) for v in k): float(x) for k, x in zip(gold["keys"], gold["values"])}
    tab = problems.shell_explicit_results(d3, shape, shell_labels, dist_kw=dist_kw)
    mine_eqs = {k[0] for k in tab}
    scale = {}
    for k, v in ref.items():
        scale[k[0]] = max(scale.get(k[0], 0.0), abs(v))
    worst, checked = 0.0, 0
    for k, v in ref.items():
        if k[0] not in mine_eqs:
            continue                                  # (a constant right-hand side: a number here, not an expression)
        err = abs(tab.get(k, 0.0) - v) / scale[k[0]]
        worst = max(worst, err)
        checked += 1
        assert err < tol, (k, tab.get(k), v)
    for k, v in tab.items():
        if k not in ref:
            assert abs(v) < tol * scale.get(k[0], 1.0), ("mode absent from the reference's table", k, v)
    assert checked > 100 and {1, 2} <= mine_eqs
    return worst
