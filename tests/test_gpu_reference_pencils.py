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


def test_benchmark_configuration_solves_the_references_matrices(full_size):
    """4 solves (2 RK222 steps) x 16 pencils of the 512 x 512 x 256 run, default kernels (one thread per system)"""
    solver, f, ref = full_size
    recs = solver.solve_probe["records"]
    assert len(recs) == 4
    summ = pencil_check.summarize(pencil_check.check_records(ref, recs, ref.groups))
    print("512x512x256 default variant:", summ)
    assert summ["pencils"] == 16
    assert summ["max_residual"] < 1e-12, summ
    assert summ["max_solution_error"] < 1e-10, summ
    assert summ["max_dropped"] < 1e-25, summ
    assert abs(float(np.sqrt(np.sum(np.asarray(f["b"]["c"]) ** 2))) - 1.0854) < 1e-3


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_every_sweep_variant_at_full_size(full_size, variant):
    solver, f, ref = full_size
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
