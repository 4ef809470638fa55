"""GPU parity (through the C ABI): ddh_regularity_recombine == the reference's ShellBasis regularity
recombination (tests/golden/shell.npz), tensor ranks 0-2, forward and backward, with and without the fused
radial factor (dR/r)^(-+k).  Tolerance rel-L2 <= 1e-14."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TAGS = ["8x6x5_k0", "16x10x6_k1"]


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "shell.npz"))


@pytest.fixture(scope="module")
def hex_():
    from dedalus_amd.executor import HipExecutor
    return HipExecutor()


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("rank", [0, 1, 2])
def test_recombination_matches_reference(gold, hex_, tag, rank):
    from dedalus_amd.core.curvilinear import RegularityRecombination
    rows = gold[tag + "__ellrows"]
    plan = RegularityRecombination(rows, tuple(int(x) for x in gold[tag + "__shape12"]), rank, executor=hex_)
    data = gold[tag + "__r%d__in" % rank]
    fac = gold[tag + "__fac1"]
    for forward, key in ((True, "fwd"), (False, "bwd")):
        ref = gold[tag + "__r%d__%s" % (rank, key)]
        d = hex_.from_host(data)
        (plan.forward if forward else plan.backward)(d)
        assert rel(hex_.download(d), ref) < 1e-14, (key,)
        # fused radial factor: forward multiplies by (dR/r)^-k before, backward by (dR/r)^k after the mixing
        f = fac ** (-1.0 if forward else 1.0)
        d = hex_.from_host(data)
        (plan.forward if forward else plan.backward)(d, hex_.from_host(f))
        assert rel(hex_.download(d), ref * f.reshape(1, 1, 1, -1)) < 1e-14, (key, "factor")


def test_recombination_large(hex_):
    """Shell-config sized slots (rank 2, 192 radial points) against the numpy oracle executor."""
    from dedalus_amd.core.curvilinear import RegularityRecombination
    from oracle.np_executor import NumpyExecutor
    rng = np.random.default_rng(3)
    n1, n2, n3 = 12, 20, 192
    rows = np.array([(l, 0, n1, l, l + 1) for l in range(n2)] + [(3, 2, 6, 5, 6)], dtype=np.int64)
    data = rng.standard_normal((9, n1, n2, n3))
    ref = data.copy()
    RegularityRecombination(rows, (n1, n2), 2, executor=NumpyExecutor()).backward(ref)
    d = hex_.from_host(data)
    RegularityRecombination(rows, (n1, n2), 2, executor=hex_).backward(d)
    assert rel(hex_.download(d), ref) < 1e-14


def test_shell_convection_config_size_sampled_ell_systems():
    """BASELINE config 5 at its real size, ShellBasis(256, 128, 128) -> Lmax 126, 384 x 192 x 192 grid: every implicit
    solve of two SBDF2 steps of the shell-convection example is compared, on sampled ell and (m, part) slots, with the
    oracle -- a direct LAPACK solve of that ell's (a M + b L) restricted to its valid modes (the reference's
    per-subproblem solve, libraries/matsolvers.py:126-149; matrices depend on ell only) -- i.e. the device
    factorization (ddh_dense_inverse_*) and the FP64 MFMA application of the inverses at config size."""
    import dedalus_amd.public as d3
    import problems
    solver, f = problems.shell_convection(d3, shape=(256, 128, 128), timestepper="SBDF2")
    assert solver.nl == 127 and solver.Nr == 128 and solver.nm == 128
    solver.solve_probe = []
    for _ in range(2):
        solver.step(0.05)
    recs, solver.solve_probe = solver.solve_probe, None
    assert len(recs) == 2
    R, nm, nl, Nr = solver.R, solver.nm, solver.nl, solver.Nr
    worst = 0.0
    for rec in recs:
        rhs = rec["rhs"].reshape(R, 2 * nm, nl, Nr)
        x = rec["x"].reshape(R, 2 * nm, nl, Nr)
        for ell in (0, 1, 2, 50, 126):
            A = rec["a"] * solver._dense(solver.M_tl, ell) + rec["b"] * solver._dense(solver.L_tl, ell)
            rv = solver.row_valid[:, ell, :].reshape(-1)
            cv = solver.col_valid[:, ell, :].reshape(-1)
            lu = np.linalg.inv(A[np.ix_(rv, cv)])
            for i1 in sorted({0, 1, 2 * min(ell, 1), 2 * ell, 2 * ell + 1}):
                if i1 // 2 > ell:
                    continue
                r, got = rhs[:, i1, ell, :].reshape(-1), x[:, i1, ell, :].reshape(-1)
                ref = np.zeros(R * Nr)
                ref[cv] = lu @ r[rv]
                if np.linalg.norm(ref) == 0.0:
                    assert np.all(got == 0.0)
                    continue
                err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
                worst = max(worst, err)
                assert err < 1e-9, (ell, i1, err)
                assert np.all(got[~cv] == 0.0)
    print("shell 256x128x128: worst per-ell solve error vs LAPACK on sampled (ell, slot):", worst)
    b = np.asarray(f["b"]["c"])
    assert np.isfinite(b).all()
