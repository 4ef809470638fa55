"""
Band structure of per-group LHS systems given as explicit matrices (the shell's per-ell systems).

The reference solves every subproblem with a sparse LU of the matrices it assembles (core/subsystems.py:481-600,
libraries/matsolvers.py:129-160).  The dense inverses this package used for the curvilinear solvers cost O(n^3) per
change of the timestep and O(n^2) storage per system; this module finds, ONCE at solver build time, the permutation /
recombination that turns a M + b L of every group into a narrow band matrix, so that the device factors it with a
partial-pivoting band LU (csrc/ddh_ellband.hip) in O(n kl (kl + ku)) and sweeps it in O(n (kl + ku)) per right-hand side:

  * boundary rows ("single-point" equations packed along n) couple every radial mode of a variable.  The columns of the
    variables they touch (components coupled by the same rows form one group, ordered n-major) are recombined,
    X = P Y with column q of P = e_q + sum_s c_s e_(q-s), so that every boundary row vanishes on the recombined columns;
    the boundary rows themselves are replaced by the equivalent combinations T = G0^-1 that pick one unknown each.
  * rows and columns are ordered n-major, every tau column placed where its first entry is (the lift taus meet the
    last rows of their equations, a constant tau the n = 0 row of its equation);
  * what remains is measured: kl / ku of the permuted matrix, and the band of P in the permuted order.

Groups whose matrices do not fit the band limits (the ell = 0 system of a shell problem with a pressure gauge: one
dense row) are left to the dense path; the caller combines both.

Everything here is setup-time host analysis on small matrices (like the assembly of the matrices themselves); the
per-timestep work -- forming a M + b L, the factorization and the sweeps -- runs on the device.
"""

import numpy as np
from scipy import sparse


class EllBandPlan:
    """Attributes (nl groups, nmax = largest system, all arrays padded):
        n[g]               system size (0: no valid modes, or a dense group)
        dense_groups       groups left to the dense path
        kl, ku, mp, nbc    band limits over the banded groups: sub / super diagonals of the permuted matrix, super
                           diagonals of P, boundary rows
        row_index[g][i]    flat (component * Nr + n) index of permuted row i (-1: padding), col_index likewise
        T[g]               (nbc, nbc) combination of the boundary rows (identity where there are none)
        P[g][i][s]         P_perm[i, i + 1 + s], s < mp
        MB, LB[g][i][d]    permuted, recombined M / L: entry (i, i - kl + d), d < kl + ku + 1
    """

    def __init__(self, M_of, L_of, row_valid, col_valid, packed_rows, packed_cols, Nr, groups, kl_max=32, w_max=96,
                 cutoff=1e-12):
        self.Nr = int(Nr)
        self.nl = len(row_valid)
        self.kl_max, self.w_max = int(kl_max), int(w_max)
        self.cutoff = cutoff
        per = {}
        self._rec_cache = {}
        self.dense_groups = []
        self.why_dense = {}
        for g in groups:
            rv, cv = np.asarray(row_valid[g]).reshape(-1), np.asarray(col_valid[g]).reshape(-1)
            if not rv.any():
                continue
            res = self._analyse(M_of(g), L_of(g), rv, cv, packed_rows, packed_cols)
            if isinstance(res, str):
                self.dense_groups.append(g)
                self.why_dense[g] = res
            else:
                per[g] = res
        # a few groups much wider than the rest (the ell = 0 system with its gauge row) would widen the windows of
        # every group: they go to the dense path as well
        if per:
            wmed = np.median([r["kl"] + r["ku"] for r in per.values()])
            for g in [g for g, r in per.items() if r["kl"] + r["ku"] > 1.5 * wmed]:
                self.why_dense[g] = "band much wider than the median (kl %d, ku %d)" % (per[g]["kl"], per[g]["ku"])
                self.dense_groups.append(g)
                del per[g]
            self.dense_groups.sort()
        self.per = per
        self._rec_cache = None
        self.n = np.zeros(self.nl, dtype=np.int32)
        if not per:
            self.kl = self.ku = self.mp = self.nbc = self.nmax = 0
            return
        self.kl = max(r["kl"] for r in per.values())
        self.ku = max(r["ku"] for r in per.values())
        self.mp = max(r["mp"] for r in per.values())
        self.nbc = max(r["T"].shape[0] for r in per.values())
        self.nmax = max(r["n"] for r in per.values())
        nl, nmax, W = self.nl, self.nmax, self.kl + self.ku + 1
        self.row_index = np.full((nl, nmax), -1, dtype=np.int64)
        self.col_index = np.full((nl, nmax), -1, dtype=np.int64)
        self.T = np.zeros((nl, max(self.nbc, 1), max(self.nbc, 1)))
        self.nbc_of = np.zeros(nl, dtype=np.int32)
        self.P = np.zeros((nl, nmax, max(self.mp, 1)))
        self.MB = np.zeros((nl, nmax, W))
        self.LB = np.zeros((nl, nmax, W))
        for g, r in per.items():
            n = r["n"]
            self.n[g] = n
            self.row_index[g, :n] = r["rows"]
            self.col_index[g, :n] = r["cols"]
            k = r["T"].shape[0]
            self.nbc_of[g] = k
            self.T[g, :k, :k] = r["T"]
            for name, dst in (("Mp", self.MB), ("Lp", self.LB)):
                i, j, v = r[name]
                dst[g, i, j - i + self.kl] = v
            i, j, v = r["Pp"]
            self.P[g, i, j - i - 1] = v

    # ------------------------------------------------------------------------------------------------------------------
    def _analyse(self, M, L, rv, cv, packed_rows, packed_cols):
        Nr = self.Nr
        ridx, cidx = np.flatnonzero(rv), np.flatnonzero(cv)
        n = len(ridx)
        if len(cidx) != n:
            return "not square"
        # (dense or scipy.sparse matrices of the full (component, n) index space; sparse from here on)
        Ms = sparse.csr_matrix(M)[ridx][:, cidx].tocsr()
        Ls = sparse.csr_matrix(L)[ridx][:, cidx].tocsr()
        rcomp, rn, ccomp, cn = ridx // Nr, ridx % Nr, cidx // Nr, cidx % Nr
        bc = np.flatnonzero(np.isin(rcomp, packed_rows))
        tau = np.flatnonzero(np.isin(ccomp, packed_cols))
        inter_r = np.flatnonzero(~np.isin(rcomp, packed_rows))
        inter_c = np.flatnonzero(~np.isin(ccomp, packed_cols))
        if len(bc) and Ms[bc].count_nonzero():
            return "boundary rows with time derivatives"
        Rb = Ls[bc].toarray() if len(bc) else np.zeros((0, n))
        if np.any(Rb[:, tau] != 0):
            return "boundary rows touch tau columns"
        # components coupled by boundary rows -> groups of variables that are recombined together
        comps = sorted(set(ccomp[inter_c].tolist()))
        parent = {c: c for c in comps}

        def find(c):
            while parent[c] != c:
                c = parent[c]
            return c
        for k in range(len(bc)):
            cs = sorted(set(ccomp[np.flatnonzero(Rb[k] != 0)].tolist()))
            for c in cs[1:]:
                parent[find(c)] = find(cs[0])
        groups = {}
        for c in comps:
            groups.setdefault(find(c), []).append(c)
        # the boundary rows (and with them the recombination) are the same for most groups: worked out once per pattern
        key = (cidx.tobytes(), bc.tobytes(), Rb.tobytes())
        rec = self._rec_cache.get(key)
        if rec is None:
            rec = self._rec_cache[key] = self._recombination(n, Rb, ccomp, cn, groups)
        if isinstance(rec, str):
            return rec
        P, T, bc_order, bc_target = rec
        Mp, Lp = (Ms @ P).tolil(), (Ls @ P).tolil()
        if len(bc):
            TL = T @ Lp[bc].toarray()
            # by construction the transformed boundary rows are unit rows: set them exactly
            unit = np.zeros((len(bc), n))
            unit[bc_order, bc_target] = 1.0
            if np.abs(TL - unit).max() > 1e-8:
                return "boundary rows do not reduce to unit rows"
            Lp[bc] = unit
        Mp, Lp = Mp.tocoo(), Lp.tocoo()
        keep = lambda A: np.abs(A.data) >= self.cutoff                 # entry_cutoff (core/subsystems.py:536)
        Mi, Mj, Mv = (a[keep(Mp)] for a in (Mp.row, Mp.col, Mp.data))
        Li, Lj, Lv = (a[keep(Lp)] for a in (Lp.row, Lp.col, Lp.data))
        # permutation: boundary rows first, then equations n-major; unknowns n-major, tau columns last
        ri = inter_r[np.lexsort((rcomp[inter_r], rn[inter_r]))]
        ci = inter_c[np.lexsort((ccomp[inter_c], cn[inter_c]))]
        # tau columns sit where their entries are: after the unknowns of the first equation row they enter (the lift
        # taus meet the last rows of their equations; a constant tau, e.g. tau_p, the n = 0 row of its equation)
        if len(tau):
            is_bc = np.zeros(n, dtype=bool)
            is_bc[bc] = True
            first_n = np.full(len(tau), Nr)
            for t, col in enumerate(tau):
                rr = np.concatenate([Mi[Mj == col], Li[Lj == col]])
                rr = rr[~is_bc[rr]]
                if len(rr):
                    first_n[t] = rn[rr].min()
            first_n = np.where(first_n < Nr // 2, first_n, Nr)          # (late ones: simply last)
            key_n = np.concatenate([cn[ci] + 0.0, first_n + 0.5])
            order = np.argsort(key_n, kind="stable")
            cols = np.concatenate([ci, tau])[order]
        else:
            cols = ci
        pos = np.empty(n, dtype=np.int64)
        pos[cols] = np.arange(n)
        order_k = sorted(range(len(bc)), key=lambda k: pos[int(bc_target[bc_order.index(k)])])
        rows = np.concatenate([bc[order_k].astype(int), ri])
        rpos = np.empty(n, dtype=np.int64)
        rpos[rows] = np.arange(n)
        Tperm = T[np.ix_(order_k, order_k)]
        Mq = (rpos[Mi], pos[Mj], Mv)
        Lq = (rpos[Li], pos[Lj], Lv)
        Pc = P.tocoo()
        off = Pc.row != Pc.col
        Pq = (pos[Pc.row[off]], pos[Pc.col[off]], Pc.data[off])
        d = np.concatenate([Mq[0] - Mq[1], Lq[0] - Lq[1]])
        kl, ku = int(max(d.max(), 0)), int(max((-d).max(), 0))
        mp = int((Pq[1] - Pq[0]).max()) if len(Pq[0]) else 0
        if len(Pq[0]) and (Pq[1] - Pq[0]).min() < 1:
            return "recombination not upper triangular in the permuted order"
        if kl > self.kl_max or kl + ku > self.w_max or mp > kl + ku:
            return "band too wide (kl %d, ku %d)" % (kl, ku)
        return dict(n=n, rows=ridx[rows], cols=cidx[cols], T=Tperm, Mp=Mq, Lp=Lq, Pp=Pq, kl=kl, ku=ku, mp=mp)

    @staticmethod
    def _recombination(n, Rb, ccomp, cn, groups):
        """-> (P sparse, T, boundary rows in the order of the unknowns they pick, those unknowns) or a reason"""
        rows_p, cols_p, vals_p = [np.arange(n)], [np.arange(n)], [np.ones(n)]
        T = np.eye(Rb.shape[0])
        bc_order, bc_target = [], []
        for _, cs in sorted(groups.items()):
            cols = np.flatnonzero(np.isin(ccomp, cs))
            cols = cols[np.lexsort((ccomp[cols], cn[cols]))]           # n-major
            touching = [k for k in range(Rb.shape[0]) if np.any(Rb[k, cols] != 0)]
            m = len(touching)
            if m == 0:
                continue
            if m > len(cols):
                return "more boundary rows than unknowns"
            Rv = Rb[np.ix_(touching, cols)]
            G0 = Rv[:, :m]
            if np.linalg.cond(G0) > 1e10:
                return "boundary rows not independent on the leading modes"
            qs = np.arange(m, len(cols))
            if len(qs):
                # column q = e_q + sum_s c_s e_(q-s): G c = -R[:, q] with G = columns q-1 ... q-m, all q at once
                Gs = Rv[:, qs[:, None] - 1 - np.arange(m)[None, :]].transpose(1, 0, 2)          # (nq, m, m)
                if np.linalg.cond(Gs).max() > 1e10:
                    return "recombination singular"
                c = np.linalg.solve(Gs, -Rv[:, qs].T[:, :, None])[:, :, 0]                       # (nq, m)
                rows_p.append(cols[qs[:, None] - 1 - np.arange(m)[None, :]].ravel())
                cols_p.append(np.repeat(cols[qs], m))
                vals_p.append(c.ravel())
            T[np.ix_(touching, touching)] = np.linalg.inv(G0)
            bc_order += touching
            bc_target += cols[:m].tolist()
        if sorted(bc_order) != list(range(Rb.shape[0])):
            return "boundary rows without unknowns"
        P = sparse.csr_matrix((np.concatenate(vals_p), (np.concatenate(rows_p), np.concatenate(cols_p))), shape=(n, n))
        return P, T, bc_order, bc_target

    # ---- host restatement of what the device does with the plan (tests) --------------------------------------------------
    def reference_solve(self, g, a, b, rhs_flat):
        """rhs_flat: (R Nr, nrhs) equation-space right-hand side of group g -> (R Nr, nrhs) solution, through the band
        path with LAPACK's gbtrf / gbtrs standing in for the device kernels."""
        from scipy.linalg import lapack
        n, kl, ku = int(self.n[g]), self.kl, self.ku
        W = kl + ku + 1
        A = a * self.MB[g, :n] + b * self.LB[g, :n]                    # [i][d] = entry (i, i - kl + d)
        ab = np.zeros((2 * kl + ku + 1, n))
        for d in range(W):
            i = np.arange(n)
            j = i - kl + d
            ok = (j >= 0) & (j < n)
            ab[kl + ku + i[ok] - j[ok], j[ok]] = A[i[ok], d]
        lu, piv, info = lapack.dgbtrf(ab, kl, ku)
        if info:
            raise np.linalg.LinAlgError("gbtrf info %d" % info)
        r = rhs_flat[self.row_index[g, :n]].astype(float)
        k = int(self.nbc_of[g])
        r[:k] = self.T[g, :k, :k] @ r[:k]
        y, info = lapack.dgbtrs(lu, kl, ku, r, piv)
        z = y.copy()
        for s in range(self.mp):
            z[:n - 1 - s] += self.P[g, :n - 1 - s, s, None] * y[1 + s:]
        out = np.zeros_like(rhs_flat, dtype=float)
        out[self.col_index[g, :n]] = z
        return out


class BandBlockPlan:
    """One explicit band system (no boundary rows, no recombination) in the attribute layout EllBandPlan hands to
    executor.EllBand: M / L given densely in the order they are to be factored in.  Used for the band block of a
    Cartesian solver's k = 0 pencil (dedalus_amd/pencilpack.py::_flagged_inverses)."""

    def __init__(self, M, L, cutoff=0.0):
        M, L = np.asarray(M, dtype=np.float64), np.asarray(L, dtype=np.float64)
        n = M.shape[0]
        i, j = np.nonzero((np.abs(M) > cutoff) | (np.abs(L) > cutoff))
        self.kl = int(max((i - j).max(), 0)) if len(i) else 0
        self.ku = int(max((j - i).max(), 0)) if len(i) else 0
        W = self.kl + self.ku + 1
        self.nl, self.nmax, self.mp, self.nbc = 1, n, 0, 0
        self.n = np.array([n], dtype=np.int32)
        self.nbc_of = np.zeros(1, dtype=np.int32)
        self.T = np.zeros((1, 1, 1))
        self.P = np.zeros((1, n, 1))
        self.MB = np.zeros((1, n, W))
        self.LB = np.zeros((1, n, W))
        self.MB[0, i, j - i + self.kl] = M[i, j]
        self.LB[0, i, j - i + self.kl] = L[i, j]
        self.row_index = self.col_index = None           # (the caller gives element offsets directly)
        self.per, self.dense_groups = {0: None}, []
