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
    """BASELINE config 5 at its real size, ShellBasis(256, 128, 128) -> Lmax 126, 384 x 192 x 192 grid, against the
    UNMODIFIED reference's subproblem matrices at that size (tests/golden/config_shell.npz: ell = 0, 1, 2, 50, 126 of
    the shell-convection example built by core/subsystems.py:497-596; matrices depend on ell only):
      * a M + b L of this package equals the reference's M_min / L_min entry by entry for every azimuthal wavenumber of
        those ell (the reference keeps the cos and msin parts of a wavenumber in one vector) and carries nothing outside
        its valid modes;
      * every implicit solve of two SBDF2 steps, gathered in the reference's order on sampled (ell, m), satisfies the
        REFERENCE's matrix and agrees with a LAPACK solve of it (libraries/matsolvers.py:126-149) -- i.e. the device
        factorization (ddh_dense_inverse_*) and the FP64 MFMA application of the inverses at config size."""
    import dedalus_amd.public as d3
    import problems
    import config_check as cc
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "config_shell.npz"))
    assert tuple(G["shape"]) == (256, 128, 128)
    solver, f = problems.shell_convection(d3, shape=(256, 128, 128), timestepper="SBDF2")
    assert solver.nl == 127 and solver.Nr == 128 and solver.nm == 128
    Tin, Tout = cc.shell_tags(solver, d3)
    solver.solve_probe = []
    for _ in range(2):
        solver.step(0.05)
    recs, solver.solve_probe = solver.solve_probe, None
    assert len(recs) == 2
    worst_m, worst_r, worst_x = 0.0, 0.0, 0.0
    for ell in (int(v) for v in G["ells"]):
        tag = "ell%d__" % ell
        ncol = G[tag + "in_var"].shape[1]
        assert ncol == ell + 1                                  # one subsystem per azimuthal wavenumber m <= ell
        seen = set()
        for col in sorted({0, 1 % ncol, ncol // 2, ncol - 1}):
            for rec in recs:
                A, slots, m, ign = cc.shell_group(solver, ell, rec["a"], rec["b"], Tin, G, tag, col)
                err, maps = cc.compare_group(A, slots, slots, Tin, Tout, G, tag, rec["a"], rec["b"], col=col, ignore=ign,
                                             want_maps=True)
                assert err < 1e-12, (ell, m, err)
                res, dx = cc.check_solve(maps, np.asarray(rec["rhs"]), np.asarray(rec["x"]))
                worst_m, worst_r, worst_x = max(worst_m, err), max(worst_r, res), max(worst_x, dx)
                assert res < 1e-10 and dx < 1e-8, (ell, m, res, dx)
            seen.add(m)
        assert len(seen) == len({0, 1 % ncol, ncol // 2, ncol - 1})
    print("shell 256x128x128 vs reference: matrices %.1e, solve residual %.1e, solution vs LAPACK %.1e"
          % (worst_m, worst_r, worst_x))
    b = np.asarray(f["b"]["c"])
    assert np.isfinite(b).all()
    # ---- the END STATE of those two steps against the unmodified reference's at this size (tests/golden/
    # config_shell_endstate.npz <- oracle/make_golden_config.py::config_shell_endstate: 25 min of host time to build the
    # reference's 127 subproblems, 38 s per step): norms of the whole arrays and every 32nd row of the packed azimuthal axis
    E = np.load(os.path.join(os.path.dirname(__file__), "golden", "config_shell_endstate.npz"))
    assert tuple(E["shape"]) == (256, 128, 128) and int(E["steps"]) == 2 and float(E["dt"]) == 0.05
    stride = int(E["sub_stride"])
    # rel-L2; the tau amplitudes are 1e-5 ... 1e-10 of the fields they correct: measured against 1e-6 where smaller
    tol = dict(b=1e-9, p=1e-6, u=1e-7, tau_p=1e-6, tau_b1=1e-5, tau_b2=1e-5, tau_u1=1e-5, tau_u2=1e-5)
    errs = {}
    for k, fld in f.items():
        c = np.asarray(fld["c"])
        ref = E["end__%s_sub" % k]
        if k == "tau_p":                                        # a constant: one number (~1e-20 here)
            errs[k] = abs(float(c.reshape(-1)[0]) - float(ref.reshape(-1)[0]))
            continue
        sub = c[..., ::stride, :, :] if (c.ndim >= 3 and c.shape[-3] >= 32) else c
        assert sub.shape == ref.shape, (k, sub.shape, ref.shape)
        floor = 1e-6 if k.startswith("tau") else 1e-300
        errs[k] = float(np.linalg.norm(sub - ref) / max(np.linalg.norm(ref), floor))
        assert abs(np.linalg.norm(c) - float(E["end__%s_norm" % k])) <= 10 * tol[k] * max(float(E["end__%s_norm" % k]), floor), k
    print("shell 256x128x128 end state vs reference:", {k: "%.1e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v < tol[k], (k, v)


@pytest.mark.gpu
def test_multistep_step_graphs_reproduce_ordinary_steps():
    """SBDF2 with a fixed timestep replayed from HIP graphs (one per phase of the history-buffer rotation,
    core/ivp_common.py::_graph_replay) against ordinary launches; the constant tau_p stays on the device between steps
    and is fetched on demand"""
    import problems
    import dedalus_amd.public as d3
    out = []
    for graph in (False, True):
        s, f = problems.shell_convection(d3, shape=(16, 8, 16), timestepper="SBDF2")
        if graph:
            s.enable_step_graph(True)
        for _ in range(12):
            s.step(0.02)
        if graph:
            assert len(s._graph["graphs"]) == 2 and not s._graph["failed"]       # both phases captured and replayed
        out.append({v.name: np.array(v["c"]) for v in s.state})
        s.step(0.01)                                                             # a new timestep: ordinary launches again
        s.step(0.01)
        out[-1]["after"] = np.array(f["b"]["c"])
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]), k
