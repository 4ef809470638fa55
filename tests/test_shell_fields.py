"""Field-level transforms of ShellBasis tensor fields (row a13) on the numpy oracle executor against the
reference's own `field['g']` / `field['c']` (tests/golden/shellfields.npz): ranks 0-2, k = 0 and k = 1 shells,
scales 1 and 3/2."""
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shellfields.npz"))
CASES = [((16, 12, 6), 0), ((8, 8, 5), 0), ((16, 10, 6), 1)]


def run_case(executor, shape, k, rank):
    import dedalus_amd.public as d3
    tag = "%dx%dx%d_k%d__" % (shape + (k,))
    coords = d3.SphericalCoordinates("phi", "theta", "r")
    dist = d3.Distributor(coords, dtype=np.float64, **({"executor": executor} if executor is not None else {}))
    shell = d3.ShellBasis(coords, shape=shape, radii=tuple(GOLD[tag + "radii"]), dealias=3 / 2, dtype=np.float64, k=k)
    f = dist.TensorField((coords,) * rank, bases=shell) if rank else dist.Field(bases=shell)
    errs = {}
    phi, theta, r = dist.local_grids(shell)
    errs["r_grid"] = np.abs(np.ravel(r) - GOLD[tag + "r_grid"]).max()
    f['c'] = GOLD[tag + "r%d__cin" % rank]
    errs["g1"] = np.abs(f['g'] - GOLD[tag + "r%d__g1" % rank]).max() / np.abs(GOLD[tag + "r%d__g1" % rank]).max()
    f.change_scales(3 / 2)
    errs["g15"] = np.abs(f['g'] - GOLD[tag + "r%d__g15" % rank]).max() / np.abs(GOLD[tag + "r%d__g15" % rank]).max()
    f.change_scales(1)
    f['g'] = GOLD[tag + "r%d__gin" % rank]
    errs["cout"] = np.abs(f['c'] - GOLD[tag + "r%d__cout" % rank]).max() / np.abs(GOLD[tag + "r%d__cout" % rank]).max()
    return errs


@pytest.mark.parametrize("shape,k", CASES)
@pytest.mark.parametrize("rank", [0, 1, 2])
def test_shell_field_transforms_oracle(shape, k, rank):
    from oracle.np_executor import NumpyExecutor
    errs = run_case(NumpyExecutor(), shape, k, rank)
    assert all(v < 1e-11 for v in errs.values()), errs


@pytest.mark.gpu
@pytest.mark.parametrize("shape,k", CASES)
@pytest.mark.parametrize("rank", [0, 1, 2])
def test_shell_field_transforms_gpu(shape, k, rank):
    errs = run_case(None, shape, k, rank)
    assert all(v < 1e-11 for v in errs.values()), errs


def _np_kw():
    from oracle.np_executor import NumpyExecutor
    return dict(executor=NumpyExecutor())


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def check_operators(dist_kw):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import problems
    import dedalus_amd.public as d3
    res = problems.shell_operator_results(d3, dist_kw=dist_kw)
    bad = {}
    for k, v in res.items():
        ref = GOLD["shellops__" + k]
        assert v.shape == ref.shape, (k, v.shape, ref.shape)
        e = np.abs(v - ref).max() / max(np.abs(ref).max(), 1e-300)
        if e > 1e-10:
            bad[k] = e
    assert not bad, bad


def check_heat(ts, dist_kw):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import problems
    import dedalus_amd.public as d3
    solver, res = problems.run_shell_heat(d3, steps=5, timestepper=ts, dist_kw=dist_kw)
    for k, v in res.items():
        ref = GOLD["heat_%s__%s" % (ts, k)]
        tol = 1e-6 if k.startswith("tau") else 1e-10
        assert _rel(v, ref) < tol, (k, _rel(v, ref))
    return solver


def test_shell_operators_oracle():
    """lap, grad, div(grad), radial interpolation of a shell scalar == the reference's evaluate()"""
    check_operators(_np_kw())


@pytest.mark.parametrize("ts", ["SBDF2", "RK222"])
def test_shell_heat_ivp_oracle(ts):
    """heat equation in the shell (tau lifts, Dirichlet rows, per-ell solves) == the reference after 5 steps"""
    check_heat(ts, _np_kw())


@pytest.mark.gpu
def test_shell_operators_gpu():
    check_operators(None)


@pytest.mark.gpu
@pytest.mark.parametrize("ts", ["SBDF2", "RK222"])
def test_shell_heat_ivp_gpu(ts):
    solver = check_heat(ts, None)
    assert solver.ex.name == "hip"


CONV_TOL = {"p": 1e-11, "b": 1e-11, "u": 1e-9, "tau_b1": 1e-8, "tau_b2": 1e-8, "tau_u1": 1e-8, "tau_u2": 1e-8}


def check_convection(ts, dist_kw):
    """Boussinesq convection in the shell (the reference's examples/ivp_shell_convection script at 16 x 12 x 8, fixed
    timestep): every piece of BASELINE config 5 -- vector/tensor transforms, radial NCC products on the LHS, tau lifts,
    vector interpolation rows, trace, pressure gauge, grid-space advection terms, the per-ell solves and the
    Hermitian-symmetry round trips -- against the reference's state after 4 steps."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import problems
    import dedalus_amd.public as d3
    solver, res = problems.run_shell_convection(d3, steps=4, timestepper=ts, dist_kw=dist_kw)
    for k, tol in CONV_TOL.items():
        ref = GOLD["conv_%s__%s" % (ts, k)]
        assert res[k].shape == ref.shape
        assert _rel(res[k], ref) < tol, (k, _rel(res[k], ref))
    assert abs(float(res["tau_p"].reshape(-1)[0])) < 1e-10
    return solver


@pytest.mark.parametrize("ts", ["SBDF2", "RK222"])
def test_shell_convection_oracle(ts):
    check_convection(ts, _np_kw())


@pytest.mark.gpu
@pytest.mark.parametrize("ts", ["SBDF2", "RK222"])
def test_shell_convection_gpu(ts):
    solver = check_convection(ts, None)
    assert solver.ex.name == "hip"


def check_cfl(dist_kw):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import problems
    import dedalus_amd.public as d3
    solver, dts, speeds, res = problems.run_shell_cfl_case(d3, dist_kw=dist_kw)
    assert np.allclose(dts, GOLD["shellcfl__dts"], rtol=1e-10, atol=0), (dts, GOLD["shellcfl__dts"])
    assert np.allclose(speeds, GOLD["shellcfl__speeds"], rtol=1e-10, atol=0)
    for k in ("p", "b", "u"):
        assert _rel(res[k], GOLD["shellcfl__" + k]) < 1e-9, (k, _rel(res[k], GOLD["shellcfl__" + k]))


def test_shell_cfl_timestep_sequence_oracle():
    """The example's adaptive loop: spherical CFL frequency (device reduction), scheduling, refactorization on every
    dt change, GlobalFlowProperty of np.sqrt(u@u) -- dt sequence and end state of the reference."""
    check_cfl(_np_kw())


@pytest.mark.gpu
def test_shell_cfl_timestep_sequence_gpu():
    check_cfl(None)


def check_analysis(dist_kw):
    """Output tasks of the shell example: er @ (-kappa grad(b) + u b) on the grid (radial NCC re-sampled on the
    dealiased grid), f(r=...), f(phi=...) at the dealias scales -- against the reference's evaluation."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import problems
    import dedalus_amd.public as d3
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shellanalysis.npz"))
    res = problems.shell_analysis_results(d3, dist_kw=dist_kw)
    for k in gold.files:
        assert res[k].shape == gold[k].shape, (k, res[k].shape, gold[k].shape)
        assert _rel(res[k], gold[k]) < 1e-11, (k, _rel(res[k], gold[k]))


def test_shell_analysis_tasks_oracle():
    check_analysis(_np_kw())


@pytest.mark.gpu
def test_shell_analysis_tasks_gpu():
    check_analysis(None)


def test_shell_radial_transform_as_gemm_oracle(monkeypatch):
    """The radial transforms as one GEMM with the shared JacobiMMT matrix (the path taken when the radial grid size
    tiles by 64, e.g. config 5), forced here at the golden sizes: same end state of the convection run."""
    monkeypatch.setenv("DDH_RADIAL_GEMM", "2")
    check_convection("SBDF2", _np_kw())
    check_operators(_np_kw())


@pytest.mark.gpu
def test_shell_radial_transform_as_gemm_gpu(monkeypatch):
    monkeypatch.setenv("DDH_RADIAL_GEMM", "2")
    solver = check_convection("SBDF2", None)
    assert solver.ex.name == "hip"
    check_operators(None)
    check_analysis(None)
