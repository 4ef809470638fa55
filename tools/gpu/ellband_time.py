"""Device time of the shell's LHS solve and factorization at the H configuration (ShellBasis(256,128,128)):
band LU (default) against the dense inverses (DDH_SHELL_DENSE=1)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import problems  # noqa: E402
import dedalus_amd.public as d3  # noqa: E402

shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (256, 128, 128)
t0 = time.time()
s, _ = problems.shell_convection(d3, shape=shape)
t1 = time.time()
lu = s.factor(1.0, 0.05 * 2 / 3)
s.ex.sync()
print("build %.1f s, first factorization (incl. band plan) %.1f s, band=%s" % (t1 - t0, time.time() - t1, bool(s._band)))
if s._band:
    pl = s._band["plan"]
    print("plan: kl %d ku %d mp %d nbc %d nmax %d dense groups %s" % (pl.kl, pl.ku, pl.mp, pl.nbc, pl.nmax, pl.dense_groups),
          s._band["dev"].info())
rhs = s.ex.zeros((s.R, s.nx, s.ny))
rhs.normal_()
x = s.ex.zeros((s.R, s.nx, s.ny))


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print("solve   %.3f ms" % timed(lambda: s.solve(lu, rhs, x), 20))
k = [0]


def refactor():
    k[0] += 1
    s.factor(1.0, 0.05 * 2 / 3 * (1 + 0.01 * k[0]), reuse=lu)


print("factor  %.3f ms" % timed(refactor, 5))
