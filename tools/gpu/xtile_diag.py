"""Diagnostic: 3-D RB end states after two RK222 steps with the state vector / the right-hand sides tile-major or natural."""
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(out):
    import problems
    import dedalus_amd.public as d3
    n = int(os.environ.get("XT_N", 256))
    solver, f = problems.rayleigh_benard_3d(d3, Nx=n, Ny=n, Nz=n, timestepper="RK222")
    for _ in range(2):
        solver.step(1e-3)
    np.savez(out, x_tiled=solver.x_tiled, rhs_tiled=int(bool(solver.timestepper._tiled)),
             **{k: np.array(f[k]["c"]) for k in ("p", "b", "u")})


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(sys.argv[1])
        sys.exit(0)
    res = {}
    for xt in (0, 1, 2):
        for rt in (0, 1):
            for perm in (0,):
                env = dict(os.environ, DDH_X_TILED=str(xt))
                if not rt:
                    env["DDH_NO_RHS_TILING"] = "1"
                out = "/tmp/xt_%d%d%d.npz" % (xt, rt, perm)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), out], env=env, capture_output=True, text=True)
                if r.returncode:
                    print(xt, rt, perm, "FAILED", r.stderr[-1500:])
                    continue
                res[(xt, rt, perm)] = dict(np.load(out))
    base = res[(0, 0, 0)]
    for key, d in res.items():
        print("x_tiled %d rhs_tiled %d perm %d (solver says %d %d):" % (key + (int(d["x_tiled"] > 0), int(d["rhs_tiled"]))),
              {k: float(np.abs(d[k] - base[k]).max()) for k in ("p", "b", "u")},
              {k: float(np.abs(base[k]).max()) for k in ("p", "b", "u")})
