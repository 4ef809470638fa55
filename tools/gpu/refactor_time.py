"""Cost of a timestep change of the headline problem (3-D Rayleigh-Benard 512 x 512 x 256, RK222): every pencil's LHS is
re-formed and re-factored on the device.  Prints the step time at a fixed timestep, the extra time of a step that changes
the timestep (min of 4), and the end-state checksum (DDH_FACTOR_ROWS=2 / 3 must agree bit for bit)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
import dedalus_amd.public as d3  # noqa: E402

size = [int(v) for v in os.environ.get("BENCH_SIZE", "512,512,256").split(",")]
solver, f = problems.rayleigh_benard_3d(d3, Nx=size[0], Ny=size[1], Nz=size[2], timestepper="RK222")
for _ in range(4):
    solver.step(1e-3)
solver.ex.sync()
t0 = time.time()
for _ in range(10):
    solver.step(1e-3)
solver.ex.sync()
step = (time.time() - t0) / 10
extra = []
for k in range(4):
    solver.ex.sync()
    t0 = time.time()
    solver.step(1e-3 * (1.0 + 0.01 * (k + 1)))
    solver.ex.sync()
    extra.append(time.time() - t0 - step)
chk = float(np.sqrt(np.sum(np.asarray(f["b"]["c"]) ** 2)))
print("FACTOR_ROWS=%s step %.2f ms, timestep change +%.2f ms (min of 4: %s), |b_c| = %.15g" % (
    os.environ.get("DDH_FACTOR_ROWS", "default"), 1e3 * step, 1e3 * min(extra), ["%.1f" % (1e3 * e) for e in extra], chk), flush=True)
