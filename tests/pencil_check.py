"""Check of the implicit solve against the REFERENCE's own pencil matrices (test infrastructure; used by the CPU and
GPU tests and by bench.py's `parity` block).

tests/golden/pencils_nz256.npz holds, for a 4 x 4 sample of wavenumber pairs of the 512 x 512 x 256 Rayleigh-Benard
problem (modes 0, 85, 170, 255 on either axis), the matrices M_min, L_min the unmodified reference builds for those
pencils (Subproblem.build_matrices, core/subsystems.py:497-596) and its pre_left / pre_right_pinv selections
(oracle/make_golden.py::golden_pencils).  A solve of the product path, x = (a M + b L)^-1 rhs, is checked on those
pencils by
    residual   |(a M_min + b L_min) pre_right_pinv x - pre_left rhs| / |pre_left rhs|
    solution   |pre_right_pinv x - SuperLU(a M_min + b L_min)^-1 pre_left rhs| / |...|   (the reference's solver,
               libraries/matsolvers.py:126-149)
with x, rhs gathered in the reference's order by SolverBase.gather_pencil."""
import os

import numpy as np
from scipy import sparse
from scipy.sparse import linalg as spla

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pencils_nz256.npz")


class ReferencePencils:
    def __init__(self, path=GOLDEN):
        d = np.load(path)
        self.stride, self.nz = [int(v) for v in d["stride"]]
        self.groups = [tuple(int(x) for x in g) for g in d["groups"]]
        self.data = {}
        for g in self.groups:
            tag = "g%d_%d__" % g
            mats = {}
            for name in ("M_min", "L_min"):
                ip = d[tag + name + "_indptr"]
                n = len(ip) - 1
                mats[name] = sparse.csr_matrix((d[tag + name + "_data"], d[tag + name + "_indices"], ip), shape=(n, n))
            mats["pre_left"] = d[tag + "pre_left_cols"]
            mats["pre_right_pinv"] = d[tag + "pre_right_pinv_cols"]
            mats["n_in"] = int(d[tag + "pre_right_pinv_ncols"])
            mats["n_out"] = int(d[tag + "pre_left_ncols"])
            self.data[g] = mats

    def modes(self, g):
        """(mx, my) mode-group indices of sample g in the problem with Lx = Ly = 4"""
        return g[0] * self.stride, g[1] * self.stride

    def check(self, g, a, b, rhs_gathered, x_gathered, solve=True, skipped=None):
        """skipped (bool per gathered unknown, or None): unknowns the solve did NOT store (intermediate Runge-Kutta stages
        leave out what nothing reads before the last stage, ddh_pencil_solve_recombined_sparse): the stored ones are
        compared with the reference's SuperLU solution, and the residual is that of the stored unknowns completed by the
        reference's values for the others."""
        m = self.data[g]
        assert rhs_gathered.size == m["n_out"], (g, rhs_gathered.size, m["n_out"])
        assert x_gathered.size == m["n_in"], (g, x_gathered.size, m["n_in"])
        rhs = rhs_gathered[m["pre_left"]]
        x = x_gathered[m["pre_right_pinv"]]
        A = (a * m["M_min"] + b * m["L_min"]).tocsc()
        drop = np.ones(x_gathered.size, dtype=bool)
        drop[m["pre_right_pinv"]] = False
        if skipped is not None and np.any(skipped):
            sk = np.asarray(skipped, dtype=bool)[m["pre_right_pinv"]]
            xr = spla.splu(A).solve(rhs)
            xh = np.where(sk, xr, x)
            out = dict(group=g, residual=float(np.linalg.norm(A @ xh - rhs) / max(np.linalg.norm(rhs), 1e-300)),
                       skipped_unknowns=int(sk.sum()))
            keep_drop = drop & ~np.asarray(skipped, dtype=bool)
            out["dropped_max"] = float(np.abs(x_gathered[keep_drop]).max()) if keep_drop.any() else 0.0
            out["solution"] = float(np.linalg.norm((x - xr)[~sk]) / max(np.linalg.norm(xr[~sk]), 1e-300))
            out["superlu_residual"] = float(np.linalg.norm(A @ xr - rhs) / max(np.linalg.norm(rhs), 1e-300))
            return out
        out = dict(group=g, residual=float(np.linalg.norm(A @ x - rhs) / max(np.linalg.norm(rhs), 1e-300)))
        # modes the reference does not keep must be empty in the product's vectors
        out["dropped_max"] = float(np.abs(x_gathered[drop]).max()) if drop.any() else 0.0
        if solve:
            xr = spla.splu(A).solve(rhs)
            out["solution"] = float(np.linalg.norm(x - xr) / max(np.linalg.norm(xr), 1e-300))
            out["superlu_residual"] = float(np.linalg.norm(A @ xr - rhs) / max(np.linalg.norm(rhs), 1e-300))
        return out


def check_records(ref, records, groups, solve=True):
    """records: solver.solve_probe['records'] -> flat list of per-(solve, pencil) results"""
    res = []
    for rec in records:
        sk = rec.get("skipped") or [None] * len(groups)
        for g, r, x, k in zip(groups, rec["rhs"], rec["x"], sk):
            out = ref.check(g, rec["a"], rec["b"], r, x, solve=solve, skipped=k)
            out["path"] = rec.get("path")
            res.append(out)
    return res


def summarize(results):
    return dict(pencils=len({r["group"] for r in results}), solves=len(results),
                max_residual=max(r["residual"] for r in results),
                max_solution_error=max((r.get("solution", 0.0) for r in results), default=None),
                max_dropped=max(r["dropped_max"] for r in results))
