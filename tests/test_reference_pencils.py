"""CPU: the oracle executor's implicit solve, the system-vector layout and SolverBase.gather_pencil against the
reference's own pencil matrices at the BASELINE coupled size (Nz = 256): the small problem the goldens were made from
(Lx = Ly = 4/85, 4 x 4 mode groups) is built with this package, stepped, and every solve of every pencil is checked
with the reference's M_min / L_min (tests/pencil_check.py).  The GPU test of the same name family runs the check at
512 x 512 x 256 (tests/test_gpu_reference_pencils.py)."""
import numpy as np

import pencil_check
import problems


def test_oracle_executor_solves_the_references_matrices():
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    ref = pencil_check.ReferencePencils()
    s, f = problems.rayleigh_benard_3d(d3, Nx=8, Ny=8, Nz=ref.nz, Lx=4 / ref.stride, Ly=4 / ref.stride,
                                       dist_kw=dict(executor=NumpyExecutor()))
    s.solve_probe = dict(groups=ref.groups, records=[])
    s.step(1e-3)
    recs = s.solve_probe["records"]
    assert len(recs) == 2 and abs(recs[0]["b"] - 1e-3 * (1 - np.sqrt(0.5))) < 1e-18
    res = pencil_check.check_records(ref, recs, ref.groups)
    summ = pencil_check.summarize(res)
    assert summ["pencils"] == 16
    assert summ["max_residual"] < 1e-13, summ
    assert summ["max_solution_error"] < 1e-11, summ
    assert summ["max_dropped"] == 0.0, summ
