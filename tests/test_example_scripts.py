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
