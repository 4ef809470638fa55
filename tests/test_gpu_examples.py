"""GPU: the reference's own example scripts -- the VERBATIM files, vendored as fixtures under tests/golden/examples/ by
oracle/make_golden_examples.py (sha256 in MANIFEST.json) -- run UNMODIFIED through dedalus_amd.compat on the MI355X
(real HIP executor, nothing injected) and reproduce the end state the UNMODIFIED reference reaches with the same file
after the same number of main-loop iterations (tests/golden/examples.npz): BASELINE configs 1, 2, 4, 5 -- adaptive-CFL
loops, flow properties and analysis handlers included.  (tests/test_example_scripts.py is the CPU twin on the oracle
executor; tests/problems.py restates the scripts with parameters for the other tests.)"""
import hashlib
import json
import os
import runpy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
EXDIR = os.path.join(HERE, "golden", "examples")


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "examples.npz"))


def run_example(name, solver_cls, monkeypatch, tmp_path):
    """runpy of the vendored file with `import dedalus.public as d3` resolving to this package; the main loop stops at
    the iteration the reference's golden was taken at."""
    import matplotlib
    matplotlib.use("Agg")
    import dedalus_amd.compat as compat
    compat.install()
    manifest = json.load(open(os.path.join(EXDIR, "MANIFEST.json")))[name]
    path = os.path.join(EXDIR, name + ".py")
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == manifest["sha256"], "vendored script was edited"
    K = int(manifest["iterations"])
    orig = solver_cls.proceed
    monkeypatch.setattr(solver_cls, "proceed", property(lambda self: orig.fget(self) and self.iteration < K))
    monkeypatch.chdir(tmp_path)
    ns = runpy.run_path(path, run_name="__main__")
    assert ns["solver"].iteration == K
    assert ns["solver"].ex.name == "hip"
    return ns


def test_kdv_burgers_script(gold, monkeypatch, tmp_path):
    from dedalus_amd.core import solvers
    ns = run_example("kdv_burgers", solvers.InitialValueSolver, monkeypatch, tmp_path)
    assert abs(ns["solver"].sim_time - float(gold["kdv_burgers__sim_time"])) < 1e-12
    ns["u"].change_scales(1)
    assert rel(ns["u"]["c"], gold["kdv_burgers__u_c"]) < 1e-10


def test_2d_rayleigh_benard_script(gold, monkeypatch, tmp_path):
    """256 x 64, RK222, d3.CFL-driven timestep, flow property, snapshot handler: 25 iterations"""
    from dedalus_amd.core import solvers
    ns = run_example("rayleigh_benard", solvers.InitialValueSolver, monkeypatch, tmp_path)
    assert abs(ns["solver"].sim_time - float(gold["rayleigh_benard__sim_time"])) < 1e-12 * max(1.0, ns["solver"].sim_time)
    for k, tol in (("b", 1e-10), ("p", 1e-9), ("u", 1e-9)):
        ns[k].change_scales(1)
        assert rel(ns[k]["c"], gold["rayleigh_benard__%s_c" % k]) < tol, k
    assert os.path.isdir(os.path.join(tmp_path, "snapshots"))


def test_sphere_shallow_water_script(gold, monkeypatch, tmp_path):
    """256 x 128: balanced-height LBVP, then 3 RK222 steps"""
    from dedalus_amd.core import sphere
    ns = run_example("shallow_water", sphere.SphereInitialValueSolver, monkeypatch, tmp_path)
    assert abs(ns["solver"].sim_time - float(gold["shallow_water__sim_time"])) < 1e-12
    for k, tol in (("h", 1e-9), ("u", 1e-9)):
        ns[k].change_scales(1)
        assert rel(ns[k]["g"], gold["shallow_water__%s_g" % k]) < tol, k


def test_shell_convection_script(gold, monkeypatch, tmp_path):
    """192 x 96 x 6, SBDF2 with the example's CFL loop, flow properties and three file handlers: 3 iterations"""
    from dedalus_amd.core import shell
    ns = run_example("shell_convection", shell.ShellInitialValueSolver, monkeypatch, tmp_path)
    assert abs(ns["solver"].sim_time - float(gold["shell_convection__sim_time"])) < 1e-12
    for k, tol in (("b", 1e-9), ("p", 1e-8), ("u", 1e-8)):
        ns[k].change_scales(1)
        assert rel(ns[k]["g"], gold["shell_convection__%s_g" % k]) < tol, k
