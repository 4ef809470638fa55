"""CPU: the oracle executor's F evaluation == the reference's, at sizes other than the golden's (embedding property)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import problems  # noqa: E402
import explicit_check  # noqa: E402


@pytest.mark.parametrize("shape", [(16, 16, 16), (24, 20, 18)])
def test_explicit_half_matches_the_reference_at_any_resolution(shape):
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    solver, f = problems.rayleigh_benard_3d(d3, Nx=shape[0], Ny=shape[1], Nz=shape[2], timestepper="RK222",
                                            dist_kw=dict(executor=NumpyExecutor()))
    worst = explicit_check.check(solver, f, tol=1e-13)
    assert worst < 1e-13


@pytest.mark.parametrize("shape", [(32, 16, 16), (40, 24, 20)])
def test_shell_explicit_half_matches_the_reference_at_any_resolution(shape):
    """Shell convection: F of the band-limited state at two resolutions == the reference's table from a small shell"""
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    worst = explicit_check.check_shell(d3, shape, dist_kw=dict(executor=NumpyExecutor()), tol=1e-12)
    assert worst < 1e-12
