"""CPU: the host layer (equation parsing, term-list assembly, IMEX stepping) driven by the numpy
oracle executor reproduces the end states of the reference itself (tests/golden/ivp.npz, generated
by oracle/make_golden.py through oracle/refshim).  This pins oracle + host logic; the GPU tests
then compare the HIP path with the same fixtures."""
import os

import numpy as np
import pytest

import problems


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


TOL = {"p": 1e-10, "b": 1e-10, "u": 1e-9, "tau_b1": 1e-5, "tau_b2": 1e-5, "tau_u1": 1e-8, "tau_u2": 1e-8}


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ivp.npz"))


@pytest.mark.parametrize("name", ["kdv64_sbdf2", "kdv64_rk443", "rb2d_32x16_rk222", "rb2d_32x16_sbdf2",
                                  "rb3d_8x12x8_rk222", "shear2d_32x64_rk222", "shear2d_32x64_sbdf2"])
def test_oracle_executor_matches_reference(gold, name):
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    solver, res = problems.run_case(d3, name, dist_kw=dict(executor=NumpyExecutor()))
    for k, v in res.items():
        ref = gold[name + "__" + k]
        assert v.shape == ref.shape
        assert rel(v, ref) < TOL.get(k, 1e-10), (name, k, rel(v, ref))


def test_timestepper_coefficients_match_reference(golden_dir):
    from dedalus_amd.core import timesteppers as T
    g = np.load(os.path.join(golden_dir, "timesteppers.npz"))
    seqs = g["timesteps"]
    for name in ("CNAB1", "SBDF1", "CNAB2", "MCNAB2", "SBDF2", "CNLF2", "SBDF3", "SBDF4"):
        cls = T.schemes[name]
        for si, seq in enumerate(seqs):
            for it in range(5):
                a, b, c = cls.compute_coefficients(list(seq), it)
                for lab, v in (("a", a), ("b", b), ("c", c)):
                    ref = g["%s_%d_%d_%s" % (name, si, it, lab)]
                    n = min(len(ref), len(v))
                    assert np.allclose(v[:n], ref[:n], rtol=1e-12, atol=1e-12 * np.abs(ref).max()), (name, it, lab)
                    assert np.all(v[n:] == 0) and np.all(ref[n:] == 0)
    for name in ("RK111", "RK222", "RK443", "RKSMR"):
        cls = T.schemes[name]
        assert np.allclose(cls.A, g[name + "_A"], rtol=0, atol=1e-15)
        assert np.allclose(cls.H, g[name + "_H"], rtol=0, atol=1e-15)
        assert np.allclose(cls.c, g[name + "_c"], rtol=0, atol=1e-15)
    assert set(T.schemes) == {"CNAB1", "SBDF1", "CNAB2", "MCNAB2", "SBDF2", "CNLF2", "SBDF3", "SBDF4",
                              "RK111", "RK222", "RK443", "RKSMR"}


def test_product_has_no_cpu_fallback():
    """Without an explicit (test-only) executor the product insists on the HIP device."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dedalus_amd.public as d3
    from dedalus_amd.libhip import DdhError
    xcoord = d3.Coordinate('x')
    dist = d3.Distributor(xcoord, dtype=np.float64)
    xb = d3.RealFourier(xcoord, size=8, bounds=(0, 1))
    u = dist.Field(name='u', bases=xb)
    u['g'] = 1.0
    with pytest.raises(DdhError):
        u['c']


def test_cfl_timestep_sequence_matches_reference(gold):
    """Adaptive stepping (CFL frequency, scheduling, clamps, refactorization on every dt change)
    against the reference's own dt sequence and end state."""
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    solver, dts, res = problems.run_cfl_case(d3, dist_kw=dict(executor=NumpyExecutor()))
    assert np.allclose(dts, gold["cfl__dts"], rtol=1e-12, atol=0)
    for k in ("p", "b", "u"):
        assert rel(res[k], gold["cfl__" + k]) < 1e-10, k


@pytest.mark.parametrize("shape", [(64, 32), (32, 24)])
def test_poisson_lbvp_matches_reference(gold, shape):
    """examples/lbvp_2d_poisson/poisson.py (tau lifts, Dirichlet + Neumann rows, low-pass filtered forcing)."""
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    solver, fields = problems.poisson_2d(d3, Nx=shape[0], Ny=shape[1], dist_kw=dict(executor=NumpyExecutor()))
    for k, f in fields.items():
        ref = gold["poisson_%dx%d__%s" % (shape + (k,))]
        tol = 1e-10 if k in ("u", "f") else 1e-6          # the tau amplitudes are ~1e-20 (spectrally small residuals)
        assert rel(np.array(f['c']), ref) < tol, (k, rel(np.array(f['c']), ref))


def test_direct_right_hand_side_path_matches_the_gather_matvec(monkeypatch):
    """F made of fused products only (Rayleigh-Benard): the forward Jacobi transform writes the equation rows of F
    (SolverBase._plan_direct_F); with DDH_NO_DIRECT_F the products go through NLbuf and the gather mat-vec instead.
    Problems whose F has other parts (shear flow: products of derivatives of two fields + linear pieces) stay general."""
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor

    def run(case, direct):
        if direct:
            monkeypatch.delenv("DDH_NO_DIRECT_F", raising=False)
        else:
            monkeypatch.setenv("DDH_NO_DIRECT_F", "1")
        return problems.run_case(d3, case, dist_kw=dict(executor=NumpyExecutor()))

    for case in ("rb3d_8x12x8_rk222", "rb2d_32x16_sbdf2"):
        s1, r1 = run(case, True)
        s0, r0 = run(case, False)
        assert s1.F_direct is not None and s0.F_direct is None
        assert s1.NLbuf is None                       # the staging vector of the gather path is not even allocated
        for k in r1:
            assert np.linalg.norm(r1[k] - r0[k]) <= 1e-13 * max(np.linalg.norm(r0[k]), 1e-300), (case, k)
    s, _ = run("kdv64_rk443", True)
    assert s.F_direct is None


@pytest.fixture(scope="module")
def gold_schemes(golden_dir):
    return np.load(os.path.join(golden_dir, "ivp_schemes.npz"))


@pytest.mark.parametrize("name", list(problems.SCHEME_CASES))
def test_every_scheme_steps_like_the_reference(gold_schemes, name):
    """All 12 registered IMEX schemes end to end (reference: tests/test_ivp.py:20-49 steps every entry of
    timesteppers.schemes): start-up order ramps of SBDF3 / SBDF4, the leap-frog history of CNLF2, variable-step
    coefficients, against the reference's own end states (oracle/make_golden.py::golden_schemes)."""
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    solver, res = problems.run_scheme_case(d3, name, dist_kw=dict(executor=NumpyExecutor()))
    assert abs(solver.sim_time - float(gold_schemes[name + "__sim_time"])) < 1e-15
    for k, v in res.items():
        ref = gold_schemes[name + "__" + k]
        assert v.shape == ref.shape
        err = problems.scheme_error(k, v, ref)
        assert err < problems.SCHEME_TOL.get(k, 1e-10), (name, k, err)
