"""Cost of a timestep change at the benchmark size: every pencil's LHS is re-formed and re-factored (factor kernel, flagged-
pencil dense inverses), measured as (step with a new dt) - (step with the same dt)."""
import sys
import time

import numpy as np
import torch

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems                                   # noqa: E402
import dedalus_amd.public as d3                   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nz = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    solver, f = problems.rayleigh_benard_3d(d3, Nx=n, Ny=n, Nz=nz, timestepper="RK222")
    dt = 1e-3
    for _ in range(3):
        solver.step(dt)
    torch.cuda.synchronize()

    def timed(dtv):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.step(dtv)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    same = min(timed(dt) for _ in range(3))
    changes = []
    for k in range(3):
        dt *= 1.1
        changes.append(timed(dt))
        timed(dt)
    print("step %.1f ms; step with a timestep change %.1f ms -> refactorization %.1f ms (min of 3)"
          % (1e3 * same, 1e3 * min(changes), 1e3 * (min(changes) - same)))
    print("b finite:", bool(np.isfinite(np.asarray(f["b"]["c"])).all()))


if __name__ == "__main__":
    main()
