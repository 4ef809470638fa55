import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
os.environ["DDH_BLOCK_INVERSE_CHECK"] = "0"
import problems
import dedalus_amd.public as d3
s, f = problems.rayleigh_benard_2d(d3, Nx=128, Nz=64, timestepper="RK222")
s.step(1e-3)
bi = s._binv
print("binv active:", bool(bi) and bool(bi["x"]))
plan, nh = bi["plan"], bi["nh"]
lu = sorted(bi["x"])[0]
idx, x = bi["x"][lu]
a, b = s._lu_params[lu]
live = np.flatnonzero(plan.n > 0)
for g in (int(live[0]), int(live[len(live)//2])):
    band = a * plan.MB[g] + b * plan.LB[g]
    Bt = np.zeros((nh, nh))
    for d in range(band.shape[1]):
        off = d - plan.kl
        i = np.arange(max(0, -off), min(nh, nh - off))
        Bt[i, i + off] = band[i, d]
    Xg = s.ex.download(x[g])
    I = np.eye(nh)
    print("g", g, "a,b", a, b, "|Bt|max", np.abs(Bt).max(), "|X|max", np.abs(Xg).max(), "cond", np.linalg.cond(Bt))
    for name, R in (("X Bt", Xg @ Bt), ("X Bt^T", Xg @ Bt.T), ("X^T Bt", Xg.T @ Bt), ("X^T Bt^T", Xg.T @ Bt.T), ("Bt X", Bt @ Xg), ("Bt^T X", Bt.T @ Xg)):
        print("   %-9s |R - I|max = %.3e" % (name, np.abs(R - I).max()))
    ref = np.linalg.inv(Bt)
    print("   |X - inv(Bt)| %.3e  |X - inv(Bt)^T| %.3e (rel to %.3e)" % (np.abs(Xg - ref).max(), np.abs(Xg - ref.T).max(), np.abs(ref).max()))
