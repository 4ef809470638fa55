"""Fill-in statistics of the batched band LU at the 3-D Rayleigh-Benard sizes: per-row maximum (over all cells)
of the last non-zero super-diagonal of U, versus the allocated partial-pivoting fill space kl + ku."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
import dedalus_amd.public as d3  # noqa: E402
from dedalus_amd import libhip  # noqa: E402

n = [int(x) for x in os.environ.get("BENCH_SIZE", "128,128,64").split(",")]
solver, f = problems.rayleigh_benard_3d(d3, Nx=n[0], Ny=n[1], Nz=n[2], timestepper="RK222")
solver.step(1e-3)
solver.ex.sync()
ts = solver.timestepper
lu = list(ts._lus.values())[0]
nint = solver.n_interior
w = np.zeros(nint, dtype=np.int32)
libhip.call("ddh_pencil_lu_row_widths", solver.pack.handle, lu, libhip.as_ip(w))
if os.environ.get("LUW_BLOCKS"):
    # fill per block of 64 stored factorizations (DDH_LUW_BLOCK is read per call): how much tighter than the global maximum?
    pairs_all = np.minimum(9, (w + 2) // 2).mean()
    for b in [int(x) for x in os.environ["LUW_BLOCKS"].split(",")]:
        os.environ["DDH_LUW_BLOCK"] = str(b)
        wb = np.zeros(nint, dtype=np.int32)
        libhip.call("ddh_pencil_lu_row_widths", solver.pack.handle, lu, libhip.as_ip(wb))
        print("block %5d: mean width %.2f, entry pairs per row %.2f (global maximum: %.2f)" % (b, wb.mean(), np.minimum(9, (wb + 2) // 2).mean(), pairs_all))
    os.environ.pop("DDH_LUW_BLOCK")
print("n_interior", nint, "kl", solver.kl, "ku", solver.ku, "allocated width", solver.kl + solver.ku)
print("row width histogram:", dict(zip(*np.unique(w, return_counts=True))))
print("mean width %.2f of %d -> U bytes read could drop to %.0f%%" % (w.mean(), solver.kl + solver.ku,
                                                                     100 * (w.mean() + 1) / (solver.kl + solver.ku + 1)))
print("first 40 rows:", w[:40].tolist())
print("last 40 rows:", w[-40:].tolist())
