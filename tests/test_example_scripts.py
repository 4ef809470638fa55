"""CPU (build container only): the reference's own example script runs UNMODIFIED against this package
through dedalus_amd.compat (numpy oracle executor injected because there is no GPU here)."""
import os
import runpy
import sys

import numpy as np
import pytest

EXAMPLE = "/root/reference/examples/ivp_2d_rayleigh_benard/rayleigh_benard.py"


@pytest.mark.skipif(not os.path.exists(EXAMPLE), reason="reference examples only exist in the build container")
def test_reference_2d_rayleigh_benard_example_runs_unmodified(monkeypatch, tmp_path):
    import dedalus_amd.compat as compat
    from dedalus_amd.core import distributor
    from dedalus_amd.core import solvers
    from oracle.np_executor import NumpyExecutor
    compat.install()
    # no GPU in this container: give every Distributor the oracle executor (test-only injection)
    orig_init = distributor.Distributor.__init__

    def init(self, *a, **k):
        k.setdefault("executor", NumpyExecutor())
        orig_init(self, *a, **k)
    monkeypatch.setattr(distributor.Distributor, "__init__", init)
    # the script integrates to t = 50 (minutes on one core): stop it after 25 iterations
    orig_proceed = solvers.InitialValueSolver.proceed

    def proceed(self):
        return orig_proceed.fget(self) and self.iteration < 25
    monkeypatch.setattr(solvers.InitialValueSolver, "proceed", property(proceed))
    monkeypatch.chdir(tmp_path)
    ns = runpy.run_path(EXAMPLE, run_name="__main__")
    solver, b, u = ns["solver"], ns["b"], ns["u"]
    assert solver.iteration == 25
    assert np.isfinite(np.asarray(b["c"])).all() and np.isfinite(np.asarray(u["c"])).all()
    assert abs(np.sqrt(np.sum(np.asarray(b["c"]) ** 2)) - 1.0854) < 1e-3


SW_EXAMPLE = "/root/reference/examples/ivp_sphere_shallow_water/shallow_water.py"


@pytest.mark.skipif(not os.path.exists(SW_EXAMPLE), reason="reference examples only exist in the build container")
def test_reference_shallow_water_example_runs_unmodified(monkeypatch, tmp_path):
    """examples/ivp_sphere_shallow_water/shallow_water.py (256 x 128, LBVP + IVP) through dedalus_amd.compat."""
    import dedalus_amd.compat as compat
    from dedalus_amd.core import sphere
    from oracle.np_executor import NumpyExecutor
    compat.install()
    orig_init = sphere.SphereDistributor.__init__

    def init(self, *a, **k):
        k.setdefault("executor", NumpyExecutor())
        orig_init(self, *a, **k)
    monkeypatch.setattr(sphere.SphereDistributor, "__init__", init)
    orig_proceed = sphere.SphereInitialValueSolver.proceed

    def proceed(self):
        return orig_proceed.fget(self) and self.iteration < 3
    monkeypatch.setattr(sphere.SphereInitialValueSolver, "proceed", property(proceed))
    monkeypatch.chdir(tmp_path)
    ns = runpy.run_path(SW_EXAMPLE, run_name="__main__")
    solver, h, u = ns["solver"], ns["h"], ns["u"]
    assert solver.iteration == 3
    assert np.isfinite(np.asarray(h["g"])).all() and np.isfinite(np.asarray(u["g"])).all()
    # the jet is still there: max zonal velocity 80 m/s in the script's units
    assert abs(np.abs(np.asarray(u["g"])[0]).max() - 80 / 6.37122e6 * 3600) < 1e-3


POISSON_EXAMPLE = "/root/reference/examples/lbvp_2d_poisson/poisson.py"


@pytest.mark.skipif(not os.path.exists(POISSON_EXAMPLE), reason="reference examples only exist in the build container")
def test_reference_poisson_lbvp_example_runs_unmodified(monkeypatch, tmp_path):
    """examples/lbvp_2d_poisson/poisson.py (256 x 128 LBVP, plots with matplotlib) through dedalus_amd.compat."""
    import matplotlib
    matplotlib.use("Agg")
    import dedalus_amd.compat as compat
    from dedalus_amd.core import distributor
    from oracle.np_executor import NumpyExecutor
    compat.install()
    orig_init = distributor.Distributor.__init__

    def init(self, *a, **k):
        k.setdefault("executor", NumpyExecutor())
        orig_init(self, *a, **k)
    monkeypatch.setattr(distributor.Distributor, "__init__", init)
    monkeypatch.chdir(tmp_path)
    ns = runpy.run_path(POISSON_EXAMPLE, run_name="__main__")
    ug = ns["ug"]
    assert ug.shape == (256, 128) and np.isfinite(ug).all()
    # boundary conditions of the script: u(y=0) = 0.025 sin(8x)
    x = np.ravel(ns["x"])
    u, coords = ns["u"], ns["coords"]
    assert os.path.exists(os.path.join(tmp_path, "poisson.pdf"))


SHELL_EXAMPLE = "/root/reference/examples/ivp_shell_convection/shell_convection.py"


@pytest.mark.skipif(not os.path.exists(SHELL_EXAMPLE), reason="reference examples only exist in the build container")
def test_reference_shell_convection_example_runs_unmodified(monkeypatch, tmp_path):
    """examples/ivp_shell_convection/shell_convection.py (192 x 96 x 6, CFL-driven SBDF2 loop, flow properties,
    output tasks) through dedalus_amd.compat."""
    import dedalus_amd.compat as compat
    from dedalus_amd.core import shell
    from oracle.np_executor import NumpyExecutor
    compat.install()
    orig_init = shell.ShellDistributor.__init__

    def init(self, *a, **k):
        k.setdefault("executor", NumpyExecutor())
        orig_init(self, *a, **k)
    monkeypatch.setattr(shell.ShellDistributor, "__init__", init)
    orig_proceed = shell.ShellInitialValueSolver.proceed

    def proceed(self):
        return orig_proceed.fget(self) and self.iteration < 3
    monkeypatch.setattr(shell.ShellInitialValueSolver, "proceed", property(proceed))
    monkeypatch.chdir(tmp_path)
    ns = runpy.run_path(SHELL_EXAMPLE, run_name="__main__")
    solver, b, u = ns["solver"], ns["b"], ns["u"]
    assert solver.iteration == 3 and solver.sim_time == 3.0
    assert np.isfinite(np.asarray(b["g"])).all() and np.isfinite(np.asarray(u["g"])).all()


def _cartesian_example(monkeypatch, tmp_path, path, max_iter):
    import dedalus_amd.compat as compat
    from dedalus_amd.core import distributor
    from dedalus_amd.core import solvers
    from oracle.np_executor import NumpyExecutor
    compat.install()
    orig_init = distributor.Distributor.__init__

    def init(self, *a, **k):
        k.setdefault("executor", NumpyExecutor())
        orig_init(self, *a, **k)
    monkeypatch.setattr(distributor.Distributor, "__init__", init)
    orig_proceed = solvers.InitialValueSolver.proceed

    def proceed(self):
        return orig_proceed.fget(self) and self.iteration < max_iter
    monkeypatch.setattr(solvers.InitialValueSolver, "proceed", property(proceed))
    monkeypatch.chdir(tmp_path)
    import matplotlib
    matplotlib.use("Agg")
    return runpy.run_path(path, run_name="__main__")


KDV_EXAMPLE = "/root/reference/examples/ivp_1d_kdv_burgers/kdv_burgers.py"
SHEAR_EXAMPLE = "/root/reference/examples/ivp_2d_shear_flow/shear_flow.py"


@pytest.mark.skipif(not os.path.exists(KDV_EXAMPLE), reason="reference examples only exist in the build container")
def test_reference_kdv_burgers_example_runs_unmodified(monkeypatch, tmp_path):
    ns = _cartesian_example(monkeypatch, tmp_path, KDV_EXAMPLE, 60)
    solver, u = ns["solver"], ns["u"]
    assert solver.iteration == 60
    assert np.isfinite(np.asarray(u["g"])).all()
    assert os.path.exists(os.path.join(tmp_path, "kdv_burgers.pdf")) or os.path.exists(os.path.join(tmp_path, "kdv_burgers.png"))


@pytest.mark.skipif(not os.path.exists(SHEAR_EXAMPLE), reason="reference examples only exist in the build container")
def test_reference_shear_flow_example_runs_unmodified(monkeypatch, tmp_path):
    """Also exercises the example's analysis handlers (tracer, pressure, vorticity snapshots) through the HDF5 writer."""
    from dedalus_amd.tools import h5lite
    ns = _cartesian_example(monkeypatch, tmp_path, SHEAR_EXAMPLE, 5)
    solver, u, s = ns["solver"], ns["u"], ns["s"]
    assert solver.iteration == 5
    assert np.isfinite(np.asarray(u["c"])).all() and np.isfinite(np.asarray(s["c"])).all()
    r = h5lite.read(os.path.join(tmp_path, "snapshots", "snapshots_s1.h5"))
    assert sorted(r["tasks"].keys()) == ["pressure", "tracer", "vorticity"]
    assert r["tasks/vorticity"].shape[0] >= 1 and np.isfinite(r["tasks/vorticity"].read(0)).all()
