"""CPU: this package's system matrices at the BASELINE config sizes of the curvilinear problems against the UNMODIFIED
reference's own subproblem matrices (core/subsystems.py:497-596) -- tests/golden/config_sphere.npz (SphereBasis(512, 256),
per-m M_min / L_min of the shallow-water example) and config_shell.npz (ShellBasis(256, 128, 128), per-ell matrices of the
shell-convection example), made by oracle/make_golden_config.py.  Entry-by-entry equality in the reference's own order of
unknowns and equations (lined up with coefficient tags, tests/config_check.py), and nothing outside its valid modes.
The matrix DEFINITION at config size is thereby pinned on the CPU; the GPU tests (test_gpu_sphere.py, test_gpu_shell.py)
add the device factorization and every implicit solve of real steps against the same reference matrices."""
import os

import numpy as np

import config_check as cc
import problems
from oracle.np_executor import NumpyExecutor

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sphere_matrices_equal_the_references_at_512x256():
    import dedalus_amd.public as d3
    G = np.load(os.path.join(GOLD, "config_sphere.npz"))
    assert tuple(G["shape"]) == (512, 256)
    solver, fields, extra = problems.shallow_water(d3, Nphi=512, Ntheta=256, dist_kw=dict(executor=NumpyExecutor()))
    # the balanced height of the example's LBVP, solved at this size by the oracle executor
    nref = float(G["h_balanced_norm"])
    assert abs(np.linalg.norm(extra["h_balanced"]) - nref) < 1e-10 * nref
    Tin, Tout = cc.sphere_tags(solver, d3)
    worst = 0.0
    for m in (int(v) for v in G["ms"]):
        for (a, b) in ((1.0, 0.0), (0.0, 1.0), (1.0, 0.37)):
            A, slots, ign = cc.sphere_group(solver, m, a, b)
            err, n = cc.compare_group(A, slots, slots, Tin, Tout, G, "m%d__" % m, a, b, ignore=ign)
            worst = max(worst, err)
            assert err < 1e-13, (m, a, b, err)
    print("sphere 512x256: |A - A_ref| / |A_ref| <= %.1e on m = %s" % (worst, list(G["ms"])))


def test_shell_matrices_equal_the_references_at_256x128x128():
    import dedalus_amd.public as d3
    G = np.load(os.path.join(GOLD, "config_shell.npz"))
    assert tuple(G["shape"]) == (256, 128, 128)
    solver, f = problems.shell_convection(d3, shape=(256, 128, 128), timestepper="SBDF2",
                                          dist_kw=dict(executor=NumpyExecutor()))
    Tin, Tout = cc.shell_tags(solver, d3)
    worst = 0.0
    for ell in (int(v) for v in G["ells"]):
        tag = "ell%d__" % ell
        ncol = G[tag + "in_var"].shape[1]
        assert ncol == ell + 1
        for col in sorted({0, 1 % ncol, ncol // 2, ncol - 1}):
            for (a, b) in ((1.0, 0.0), (0.0, 1.0), (1.5, 0.05)):
                A, slots, m, ign = cc.shell_group(solver, ell, a, b, Tin, G, tag, col)
                err, n = cc.compare_group(A, slots, slots, Tin, Tout, G, tag, a, b, col=col, ignore=ign)
                worst = max(worst, err)
                assert err < 1e-13, (ell, m, a, b, err)
    print("shell 256x128x128: |A - A_ref| / |A_ref| <= %.1e on ell = %s" % (worst, list(G["ells"])))
