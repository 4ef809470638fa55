"""GPU parity AT THE BENCHMARKED CONFIGURATION (3-D Rayleigh-Benard 512 x 512 x 256, RK222): every sweep variant of
ddh_pencil_solve -- including the one-thread-per-system real-graded kernels bench.py runs -- against the REFERENCE's
own pencil matrices on a 4 x 4 sample of wavenumber pairs (modes 0, 85, 170, 255 per axis; tests/pencil_check.py,
tests/golden/pencils_nz256.npz), and the same kernels forced on the smaller problems whose reference end states are
committed (tests/golden/ivp_large.npz: 3-D 32^3 after 5 steps, 2-D 512 x 256 after 13 steps -- full arrays)."""
import os

import numpy as np
import pytest

import pencil_check
import problems

pytestmark = pytest.mark.gpu

# (mode, forward cooperative, backward lanes per system): the four combinations launch_solve can choose
VARIANTS = {"1+1": (0, 0, 0), "1+4": (0, 0, 4), "16+4": (0, 1, 4), "16+16": (2, 1, 16)}


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def full_size():
    import dedalus_amd.public as d3
    ref = pencil_check.ReferencePencils()
    solver, f = problems.rayleigh_benard_3d(d3, Nx=512, Ny=512, Nz=ref.nz, timestepper="RK222")
    solver.solve_probe = dict(groups=[ref.modes(g) for g in ref.groups], records=[])
    for _ in range(2):
        solver.step(1e-3)
    yield solver, f, ref
    solver.solve_probe = None


@pytest.fixture(scope="module")
def full_size_unpaired():
    """The same run with one factorization per pencil (DDH_PAIR=0): the cooperative sweep variants need it."""
    import dedalus_amd.public as d3
    ref = pencil_check.ReferencePencils()
    old = os.environ.get("DDH_PAIR")
    os.environ["DDH_PAIR"] = "0"
    try:
        solver, f = problems.rayleigh_benard_3d(d3, Nx=512, Ny=512, Nz=ref.nz, timestepper="RK222")
    finally:
        if old is None:
            del os.environ["DDH_PAIR"]
        else:
            os.environ["DDH_PAIR"] = old
    assert solver.pairing is None
    solver.solve_probe = dict(groups=[ref.modes(g) for g in ref.groups], records=[])
    for _ in range(2):
        solver.step(1e-3)
    yield solver, f, ref
    solver.solve_probe = None


def test_benchmark_configuration_solves_the_references_matrices(full_size):
    """4 solves (2 RK222 steps) x 16 pencils of the 512 x 512 x 256 run, default kernels (one thread per system)"""
    solver, f, ref = full_size
    recs = solver.solve_probe["records"]
    assert len(recs) == 4
    # the probe rides on the kernels the benchmark times: fused right-hand-side terms, structural-zero mask, and -- on the
    # intermediate stage -- the unknowns nothing reads left unstored (checked there on the stored unknowns)
    for k, rec in enumerate(recs):
        assert "ddh_pencil_solve_recombined_tiled" in rec["path"], rec["path"]      # (masked + tile-major terms)
        assert rec["terms"] == (2 if k % 2 == 0 else 4) and rec["zero_rows"]
        assert rec["skip_rows"] == (k % 2 == 0)
    summ = pencil_check.summarize(pencil_check.check_records(ref, recs, ref.groups))
    print("512x512x256 default variant:", summ)
    assert summ["pencils"] == 16
    assert summ["max_residual"] < 1e-12, summ
    assert summ["max_solution_error"] < 1e-10, summ
    assert summ["max_dropped"] < 1e-25, summ
    assert abs(float(np.sqrt(np.sum(np.asarray(f["b"]["c"]) ** 2))) - 1.0854) < 1e-3


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_every_sweep_variant_at_full_size(full_size_unpaired, full_size, variant):
    solver, f, ref = full_size_unpaired
    ts = solver.timestepper
    lu = list(ts._lus.values())[0]
    out = solver.ex.empty((solver.R, solver.nx, solver.ny))
    solver.solve_probe["records"] = []
    solver.pack.set_solve_variant(*VARIANTS[variant])
    try:
        solver.solve(lu, solver._last_rhs, out)  # the last stage's right-hand side (materialised while probing)
        solver.ex.sync()
    finally:
        solver.pack.set_solve_variant(1, -1, -1)
    summ = pencil_check.summarize(pencil_check.check_records(ref, solver.solve_probe["records"], ref.groups))
    print(variant, summ)
    assert summ["max_residual"] < 1e-12 and summ["max_solution_error"] < 1e-10, (variant, summ)
    # and the whole vector agrees with the state the default kernels produced from the same right-hand side
    assert rel(solver.ex.download(out), solver.ex.download(solver.X)) < 1e-12
    # ... and with the state of the run whose partner pencils share factorizations
    assert rel(solver.ex.download(out), solver.ex.download(full_size[0].X)) < 1e-11


def test_fused_rhs_combination_matches_materialised_rhs(full_size):
    """ddh_pencil_solve_lincomb (RHS formed inside the forward sweep) == lincomb kernel + ddh_pencil_solve, at the
    benchmark size, for the 4-term combination of the second RK222 stage"""
    solver, f, ref = full_size
    ts = solver.timestepper
    lu = list(ts._lus.values())[0]
    xs = [ts.MX0, ts.F[0], ts.F[1], ts.MX[1]]
    al = [0.3, -1.7e-3, 2.2e-3, 0.7]
    ex = solver.ex
    rhs, y1, y2 = (ex.empty((solver.R, solver.nx, solver.ny)) for _ in range(3))
    ex.lincomb(rhs, xs, al)
    solver.pack.solve(lu, rhs, y1)
    solver.pack.solve_lincomb(lu, xs, al, y2)
    ex.sync()
    assert rel(ex.download(y2), ex.download(y1)) < 1e-14


@pytest.fixture(scope="module")
def large(golden_dir):
    return np.load(os.path.join(golden_dir, "ivp_large.npz"))


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_rb3d_32_arrays_with_forced_variant(large, variant):
    import dedalus_amd.public as d3
    solver, f = problems.rayleigh_benard_3d(d3, Nx=32, Ny=32, Nz=32, timestepper="RK222")
    solver.pack.set_solve_variant(*VARIANTS[variant])
    for _ in range(5):
        solver.step(1e-3)
    for k, tol in (("p", 1e-10), ("b", 1e-10), ("u", 1e-9)):
        err = rel(np.asarray(f[k]["c"]), large["rb3d_32__" + k])
        assert err < tol, (variant, k, err)
    val = float(np.sqrt(np.sum(np.asarray(f["b"]["c"]) ** 2)))
    assert abs(val - 1.085405276538733e+00) / 1.085405276538733e+00 < 1e-12


@pytest.mark.parametrize("variant", ["1+1", "16+16"])
def test_rb2d_512x256_arrays_with_forced_variant(large, variant):
    """BASELINE config 2 (Nz = 256: the band windows of the 3-D benchmark) after 13 RK222 steps: arrays, not norms"""
    import dedalus_amd.public as d3
    solver, f = problems.rayleigh_benard_2d(d3, Nx=512, Nz=256, timestepper="RK222")
    solver.pack.set_solve_variant(*VARIANTS[variant])
    for _ in range(13):
        solver.step(1e-3)
    for k, tol in (("p", 1e-10), ("b", 1e-10), ("u", 1e-9)):
        a = np.asarray(f[k]["c"])
        err = rel(a[..., ::8, :], large["rb2d_512x256__" + k])
        assert err < tol, (variant, k, err)
        assert abs(np.linalg.norm(a) - float(large["rb2d_512x256__" + k + "_norm"])) <= 1e-10 * np.linalg.norm(a)


# ---- partner pencils: (my, mx) solved with the factorization of (mx, my) (ddh_pencil_set_pairing) ------------------------
def test_full_size_run_uses_partner_pencils(full_size):
    """The 512 x 512 x 256 configuration is symmetric in x <-> y and large enough: its factorizations are shared by
    partner cells (the sample of the tests above holds both members of the pairs (85, 170) and (170, 85) etc., the
    diagonal cells and the cells on the axes), and the factor storage is about half of one factorization per cell."""
    solver, f, ref = full_size
    assert solver.pairing is not None
    lu = list(solver.timestepper._lus.values())[0]
    per_cell = solver.pack.lu_bytes(lu) / (256 * 256)
    n, bw = solver.n_interior, solver.kl + (solver.kl + solver.ku) + 1
    assert per_cell < 0.62 * n * bw * 8 + 0.2 * n * bw * 8, per_cell


@pytest.mark.parametrize("shape", [(16, 16, 16), (24, 24, 8), (10, 10, 12)])
def test_partner_pencils_match_unpaired_solves_and_the_oracle(shape, monkeypatch):
    """Small symmetric 3-D problems with pairing forced on (DDH_PAIR_MIN=0) against pairing off (DDH_PAIR=0) and against
    the CPU oracle, which factors every pencil on its own."""
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    kw = dict(Nx=shape[0], Ny=shape[1], Nz=shape[2], timestepper="RK222")
    monkeypatch.setenv("DDH_PAIR_MIN", "0")
    s1, f1 = problems.rayleigh_benard_3d(d3, **kw)
    assert s1.pairing is not None
    monkeypatch.setenv("DDH_PAIR", "0")
    s0, f0 = problems.rayleigh_benard_3d(d3, **kw)
    assert s0.pairing is None
    monkeypatch.delenv("DDH_PAIR")
    so, fo = problems.rayleigh_benard_3d(d3, dist_kw=dict(executor=NumpyExecutor()), **kw)
    for _ in range(4):
        for s in (s1, s0, so):
            s.step(1e-3)
    for k in ("p", "b", "u"):
        a1, a0, ao = (np.array(f[k]['c']) for f in (f1, f0, fo))
        assert np.isfinite(a1).all()
        assert rel(a1, a0) < 1e-11, (k, rel(a1, a0))
        assert rel(a1, ao) < 1e-9, (k, rel(a1, ao))


def test_partner_pencils_reproduce_the_reference_end_state(large, monkeypatch):
    """rb3d 32^3, 5 steps: the reference's own end state (tests/golden/ivp_large.npz) with pairing forced on."""
    import dedalus_amd.public as d3
    monkeypatch.setenv("DDH_PAIR_MIN", "0")
    solver, f = problems.rayleigh_benard_3d(d3, Nx=32, Ny=32, Nz=32, timestepper="RK222")
    assert solver.pairing is not None
    solver.pack.set_solve_variant(0, 0, 0)
    for _ in range(5):
        solver.step(1e-3)
    for k, tol in (("p", 1e-10), ("b", 1e-10), ("u", 1e-9)):
        got, want = np.array(f[k]['c']), large["rb3d_32__" + k]
        assert rel(got, want) < tol, (k, rel(got, want))


def test_anisotropic_box_is_not_paired(monkeypatch):
    import dedalus_amd.public as d3
    monkeypatch.setenv("DDH_PAIR_MIN", "0")
    solver, f = problems.rayleigh_benard_3d(d3, Nx=8, Ny=8, Nz=8, Lx=4, Ly=2)
    assert solver.pairing is None
    solver, f = problems.rayleigh_benard_3d(d3, Nx=8, Ny=12, Nz=8)
    assert solver.pairing is None
