"""
Solvers.  InitialValueSolver keeps the reference's interface (core/solvers.py:503-806:
step / proceed / evolve / sim_time / iteration / stop_* / log_stats) while the per-pencil Python
loops of the reference (loops A and C of SURVEY.md section 3.1) are replaced by batched device
calls over ALL pencils:

  state X, M.X, L.X, F, RHS : [rows][nx][ny] system vectors in HBM (state fields are views into X)
  M, L                      : term lists shared by all pencils (no per-pencil matrices)
  LHS solves                : bordered band LU, factored on the device when (a0, b0) / k*H_ii changes
"""

import time

import os

import numpy as np

from . import operators as ops
from . import timesteppers as ts_mod
from .evaluator import Evaluator, SystemBuffer, _full_sep
from .field import Field
from .ivp_common import IVPLifecycle
from .polyop import flatten
from .problems import LinCtx

import logging
logger = logging.getLogger(__name__)


class SolverBase:
    def __init__(self, problem, **kw):
        self.problem = problem
        self.dist = dist = problem.dist
        self.ex = dist.executor
        self.variables = list(problem.variables)
        self.equations = list(problem.equations)
        self.state = self.variables
        self.evaluator_core = Evaluator(dist, self.variables)
        nf, nx, ny, kx, ky = self.evaluator_core.geom()
        self.nf, self.nx, self.ny = nf, nx, ny
        sep = dist.separable_axes

        def axes_bits(domain):
            bits = 0
            for i in range(2):
                if i >= nf or domain.by_axis[sep[i]] is not None:
                    bits |= (1 << i)
            return bits

        # ---- variable rows
        row = 0
        self.var_info = []
        for v in self.variables:
            nz = dist.coupled_size(v.domain)
            info = dict(field=v, row0=row, nz=nz, ncomp=v.ncomp, rows=v.ncomp * nz, bits=axes_bits(v.domain),
                        interior=(nz > 1), aliased=_full_sep(dist, v.domain))
            self.var_info.append(info)
            row += info["rows"]
        self.R = row
        # ---- equation rows
        row = 0
        self.eq_info = []
        for eq in self.equations:
            nz = dist.coupled_size(eq["domain"])
            info = dict(eq=eq, row0=row, nz=nz, ncomp=eq["ncomp"], rows=eq["ncomp"] * nz, bits=axes_bits(eq["domain"]),
                        interior=(nz > 1))
            self.eq_info.append(info)
            row += info["rows"]
        if row != self.R:
            raise ValueError("Problem is not square: %d equation rows vs %d variable rows (pencil-wise counts "
                             "may still differ, this is the global count)" % (row, self.R))
        # ---- state vector; state fields become views into it
        self.X = self.ex.zeros((self.R, nx, ny))
        self.x_tiled = 0                    # row length when X is kept tile-major (_enable_state_tiling: IVPs)
        self.x_banded = 0                   # its row count when, in addition, the rows of a kx band lie together
        self.sysbuf = SystemBuffer(self.X, self.R)
        for info in self.var_info:
            v = info["field"]
            old = v.coeff_data()
            if info["aliased"]:
                view = self.X[info["row0"]:info["row0"] + info["rows"]].reshape(v._storage_shape("c", None))
                self.ex.copy(view, old)
                v._c = view
                v._tiled, v._nrows = None, 0    # (a field another solver kept tile-major: its data is this solver's now)
                v._adopted = True
                v._sysbuf = self.sysbuf
                v._row0 = info["row0"]
        # ---- matrices as term lists (physical row / column numbering)
        self.M_tl = self._assemble("M")
        self.L_tl = self._assemble("L")
        self._decouple_boundary_rows()
        self.pack = self.ex.make_pack(nf, self.R, nx, ny, kx, ky, dist._mx_offset)
        self.M_id = self.pack.add_matrix(self.M_tl)
        self.L_id = self.pack.add_matrix(self.L_tl)
        self._build_recombination()
        self._build_ordering()
        self._build_F_plan()

    # ---- assembly ----------------------------------------------------------------------------------------
    def _flags(self, domain):
        sep = self.dist.separable_axes
        fx = self.nf >= 1 and domain.by_axis[sep[0]] is None
        fy = self.nf >= 2 and domain.by_axis[sep[1]] is None
        return fx, fy

    def _assemble(self, which):
        var_of = {id(i["field"]): i for i in self.var_info}
        blocks = []
        for einfo in self.eq_info:
            le = einfo["eq"][which]
            efx, efy = self._flags(einfo["eq"]["domain"])
            for leaf, terms in le.leaves.items():
                vinfo = var_of[id(leaf)]
                vfx, vfy = self._flags(leaf.domain)
                blocks.append((einfo["row0"], le.nzo, vinfo["row0"], vinfo["nz"], terms, efx or vfx, efy or vfy))
        return flatten(blocks, self.R, self.R)

    def _classify(self):
        full = 3 if self.nf >= 2 else (1 if self.nf == 1 else 0)
        def split(infos):
            inter = [i for i in infos if i["interior"]]
            edge = [i for i in infos if not i["interior"] and (i["bits"] & full) == full]
            special = [i for i in infos if not i["interior"] and (i["bits"] & full) != full]
            return inter, edge, special
        return split(self.eq_info), split(self.var_info)

    def _boundary_rows(self):
        """Rows of the boundary-condition equations (equations without a coupled basis that exist for every pencil)."""
        (eq_int, eq_bc, eq_sp), _ = self._classify()
        return [r for i in eq_bc for r in range(i["row0"], i["row0"] + i["rows"])]

    def _boundary_matrix(self, M_tl, L_tl, bc_rows):
        """Dense [len(bc_rows)][R] coefficients of the boundary rows (they must not depend on the wavenumber)."""
        R = np.zeros((len(bc_rows), self.R))
        rowpos = {r: k for k, r in enumerate(bc_rows)}
        for tl in (M_tl, L_tl):
            for t in np.flatnonzero(np.isin(tl.row, bc_rows)):
                if tl.ex[t] or tl.ey[t] or tl.dx[t] or tl.dy[t] or abs(tl.coef[t].imag) > 0:
                    raise NotImplementedError("boundary conditions that depend on the wavenumber")
                R[rowpos[int(tl.row[t])], tl.col[t]] += tl.coef[t].real
        return R

    def _recombination_matrix(self, M_tl, L_tl):
        """Column recombination X = P Y that makes the boundary-condition rows sparse (None: nothing to recombine).

        The tau-bordered pencil matrices are only well conditioned as a whole: their banded interior
        block alone is exponentially ill conditioned (it is a spectral-space shooting problem), so the
        dense boundary rows must be available as pivots.  Instead of carrying dense rows through the
        factorization we change the basis of every interior variable so that all but its first m modes
        satisfy its m boundary conditions identically,
            phi_n = p_n + sum_{s=1..m} c_{n,s} p_{n-s},    R_r[n] + sum_s c_{n,s} R_r[n-s] = 0   (r = 1..m)
        (for Dirichlet data at both ends this is the classical p_n - p_n(1)/p_{n-2}(1) p_{n-2} basis).
        The rows R.P then only touch modes < m, sit at the top of the matrix, and (a M + b L) P is a
        plain band matrix on which standard partial pivoting (LAPACK gbtrf-style) is backward stable.
        The reference reaches the same goal with SuperLU + COLAMD on each pencil
        (libraries/matsolvers.py:126-149); its `Woodbury` solver (:285-321) is the dense-border
        alternative that this replaces."""
        from scipy import sparse
        (eq_int, eq_bc, eq_sp), (var_int, var_tau, var_sp) = self._classify()
        bc_rows = self._boundary_rows()
        if not bc_rows or not var_int:
            return None
        R = self._boundary_matrix(M_tl, L_tl, bc_rows)
        P = sparse.identity(self.R, format="lil")
        used = np.zeros(len(bc_rows), dtype=bool)
        for vi in var_int:
            for c in range(vi["ncomp"]):
                cols = vi["row0"] + c * vi["nz"] + np.arange(vi["nz"])
                touching = [k for k in range(len(bc_rows)) if np.any(R[k, cols] != 0)]
                m = len(touching)
                if m == 0:
                    continue
                for k in touching:
                    others = np.delete(np.arange(self.R), cols)
                    if np.any(R[k, others] != 0):
                        raise NotImplementedError("boundary condition coupling several variables")
                    used[k] = True
                Rv = R[np.ix_(touching, cols)]
                for n in range(m, vi["nz"]):
                    G = np.array([[Rv[r, n - s1] for s1 in range(1, m + 1)] for r in range(m)])
                    cvec = np.linalg.solve(G, -Rv[:, n])
                    for s1 in range(1, m + 1):
                        if cvec[s1 - 1] != 0.0:
                            P[cols[n - s1], cols[n]] = cvec[s1 - 1]
        if not used.all():
            raise NotImplementedError("boundary condition row without interior variable")
        return sparse.csr_matrix(P)

    def _build_recombination(self):
        """X = P Y (`_recombination_matrix`) and the recombined term lists (a M + b L) P registered with the pack."""
        self.P_tl = None
        self.P_id = None
        self.MP_id, self.LP_id = self.M_id, self.L_id
        self.MP_tl, self.LP_tl = self.M_tl, self.L_tl
        P = self._recombination_matrix(self.M_tl, self.L_tl)
        if P is None:
            return
        coo = P.tocoo()
        from ..pencilpack import TermList
        self.P_tl = TermList(self.R, self.R, coo.row, coo.col, coo.data.astype(complex))
        self.MP_tl = _termlist_times_matrix(self.M_tl, P)
        self.LP_tl = _termlist_times_matrix(self.L_tl, P)
        self.P_id = self.pack.add_matrix(self.P_tl)
        self.MP_id = self.pack.add_matrix(self.MP_tl)
        self.LP_id = self.pack.add_matrix(self.LP_tl)

    # ---- independent blocks of the pencil matrix -----------------------------------------------------------------
    def _band_components(self, MP_tl, LP_tl):
        """Connected components of the bipartite row / column graph of the band block of (a M + b L) P (every term counts,
        whatever its wavenumber factor).  Returns (number of components, label per physical row, label per physical
        column); rows / columns of the small border (gauge conditions) get the label -1."""
        from scipy import sparse
        from scipy.sparse.csgraph import connected_components
        (eq_int, eq_bc, eq_sp), (var_int, var_tau, var_sp) = self._classify()
        in_rows = np.zeros(self.R, dtype=bool)
        in_cols = np.zeros(self.R, dtype=bool)
        for i in eq_int + eq_bc:
            in_rows[i["row0"]:i["row0"] + i["rows"]] = True
        for i in var_int + var_tau:
            in_cols[i["row0"]:i["row0"] + i["rows"]] = True
        row = np.concatenate([MP_tl.row, LP_tl.row]).astype(np.int64)
        col = np.concatenate([MP_tl.col, LP_tl.col]).astype(np.int64)
        keep = in_rows[row] & in_cols[col]
        row, col = row[keep], col[keep]
        A = sparse.coo_matrix((np.ones(row.size), (row, col + self.R)), shape=(2 * self.R, 2 * self.R))
        nc, lab = connected_components(A + A.T, directed=False)
        lr = np.where(in_rows, lab[:self.R], -1)
        lc = np.where(in_cols, lab[self.R:], -1)
        used = np.unique(np.concatenate([lr[lr >= 0], lc[lc >= 0]]))
        remap = -np.ones(nc + 1, dtype=np.int64)
        remap[used] = np.arange(used.size)
        return used.size, remap[lr], remap[lc]

    def _decouple_boundary_rows(self):
        """Equivalent boundary equations that let the pencil matrix fall apart into independent blocks.

        For problems that are symmetric under a reflection of the coupled axis (constant-coefficient equations between
        two plates: Rayleigh-Benard, channel flows, ...) even and odd polynomial modes only meet in the boundary
        conditions: "f(z=0) = a, f(z=Lz) = b" touch both parities.  The m boundary rows of a variable are replaced by the
        equivalent combinations T (rows) with T = G^-1, G the m x m block of their coefficients on the variable's first m
        modes: after the column recombination X = P Y each new row touches ONE mode.  If the band block of
        (a M + b L) P then has several connected components (two for the parity split), `_build_ordering` orders it
        block by block: a block-diagonal band matrix with half the bandwidth -- half the factor bytes and multiply-adds of
        every solve, and rows of different blocks are independent.  The equations solved are T (M dt X + L X) = T F:
        `self.eq_T` multiplies the boundary rows of M, L and of every right-hand-side source (`_eq_transform`,
        `F_const`); vectors in equation space are handed out un-transformed (`gather_pencil`, `equation_space_to_user`).
        Nothing changes (eq_T = None) when the split does not happen or DDH_NO_SPLIT is set.
        The reference solves the coupled matrix per pencil (core/subsystems.py:497-596, libraries/matsolvers.py:126-149)."""
        from scipy import sparse
        self.eq_T = self.eq_Tinv = None
        if os.environ.get("DDH_NO_SPLIT") is not None:
            return
        try:
            bc_rows = self._boundary_rows()
            (eq_int, eq_bc, eq_sp), (var_int, var_tau, var_sp) = self._classify()
            if not bc_rows or not var_int:
                return
            Rm = self._boundary_matrix(self.M_tl, self.L_tl, bc_rows)
        except NotImplementedError:
            return                                      # (reported by _build_recombination)
        T = np.eye(len(bc_rows))
        changed = False
        claimed = set()                                  # boundary rows already combined for another component
        for vi in var_int:
            for c in range(vi["ncomp"]):
                cols = vi["row0"] + c * vi["nz"] + np.arange(vi["nz"])
                touching = [k for k in range(len(bc_rows)) if np.any(Rm[k, cols] != 0)]
                m = len(touching)
                if m < 2 or m > vi["nz"]:
                    continue
                if claimed & set(touching):
                    # a boundary row that couples two variables (a Robin / coupled condition) belongs to two such sets: the
                    # second block would overwrite part of the first.  No equivalent rows for this problem.
                    return
                G = Rm[np.ix_(touching, cols[:m])]
                if np.linalg.cond(G) > 1e6:             # (e.g. two Neumann conditions: the constant mode drops out)
                    continue
                T[np.ix_(touching, touching)] = np.linalg.inv(G)
                claimed |= set(touching)
                changed = True
        if not changed or np.linalg.cond(T) > 1e8:
            return
        Tfull = sparse.identity(self.R, format="lil")
        for i, ri in enumerate(bc_rows):
            for j, rj in enumerate(bc_rows):
                if T[i, j] != 0.0 or i == j:
                    Tfull[ri, rj] = T[i, j]
        Tfull = sparse.csr_matrix(Tfull)
        M2, L2 = _matrix_times_termlist(Tfull, self.M_tl), _matrix_times_termlist(Tfull, self.L_tl)
        try:
            P = self._recombination_matrix(M2, L2)
        except NotImplementedError:
            return
        if P is None:
            return
        n_old = self._band_components(_termlist_times_matrix(self.M_tl, P), _termlist_times_matrix(self.L_tl, P))[0]
        n_new, lr, lc = self._band_components(_termlist_times_matrix(M2, P), _termlist_times_matrix(L2, P))
        if n_new <= n_old or any(np.sum(lr == k) != np.sum(lc == k) for k in range(n_new)):
            return
        self.eq_T = Tfull
        Tinv = sparse.identity(self.R, format="lil")
        Ti = np.linalg.inv(T)
        for i, ri in enumerate(bc_rows):
            for j, rj in enumerate(bc_rows):
                if abs(Ti[i, j]) > 1e-15 or i == j:
                    Tinv[ri, rj] = Ti[i, j]
        self.eq_Tinv = sparse.csr_matrix(Tinv)
        self._eq_T_rows = np.asarray(bc_rows)
        self.M_tl, self.L_tl = M2, L2

    def _eq_transform(self, tl):
        """T applied to the rows of a term list that maps into equation space (right-hand-side sources)."""
        return tl if self.eq_T is None else _matrix_times_termlist(self.eq_T, tl)

    def equation_space_to_user(self, vec):
        """Host copy [R][nx][ny] of a vector in equation space (F, M.X, right-hand sides) in the USER's equations: the
        boundary rows are stored as the equivalent combinations T (rows) of `_decouple_boundary_rows`."""
        v = np.array(self.ex.download(vec), dtype=float).reshape(self.R, -1)
        if self.eq_Tinv is not None:
            rows = self._eq_T_rows
            v[rows] = self.eq_Tinv[rows][:, rows] @ v[rows]
        return v.reshape(self.R, self.nx, self.ny)

    def _build_ordering(self):
        """Logical ordering for the band LU: boundary-condition rows first, then the interior equations
        with the coupled-axis index outermost; columns: interior variables (coupled index outermost),
        then the tau columns.  Rows / columns that only exist in the k=0 pencil (gauge conditions) form a
        small border.  (The reference's own orderings, core/subsystems.py:614-739, are not banded for
        real-Fourier pencils: SURVEY.md section 7.2.)"""
        (eq_int, eq_bc, eq_sp), (var_int, var_tau, var_sp) = self._classify()

        def kz_major(infos):
            nzs = {i["nz"] for i in infos}
            if len(nzs) > 1:
                raise NotImplementedError("interior blocks with different coupled sizes")
            nz = nzs.pop() if nzs else 0
            return [i["row0"] + c * i["nz"] + kz for kz in range(nz) for i in infos for c in range(i["ncomp"])]

        def flat(infos):
            return [r for i in infos for r in range(i["row0"], i["row0"] + i["rows"])]

        rows = flat(eq_bc) + kz_major(eq_int)
        cols = kz_major(var_int) + flat(var_tau)
        if len(rows) != len(cols):
            raise ValueError("band block is not square: %d equation rows vs %d variable columns "
                             "(boundary conditions and tau variables must balance)" % (len(rows), len(cols)))
        # independent blocks (connected components of the band block, `_decouple_boundary_rows`): block after block, the
        # order inside a block unchanged -- a block-diagonal band matrix whose bandwidth is that of its widest block
        self.n_blocks = 1
        if rows and os.environ.get("DDH_NO_SPLIT") is None:
            nc, lr, lc = self._band_components(self.MP_tl, self.LP_tl)
            if nc > 1 and all(np.sum(lr == k) == np.sum(lc == k) for k in range(nc)):
                rows = [r for k in range(nc) for r in rows if lr[r] == k]
                cols = [c for k in range(nc) for c in cols if lc[c] == k]
                self.n_blocks = nc
                self.block_sizes = [int(np.sum(lr == k)) for k in range(nc)]
                if len(set(self.block_sizes)) == 1 and hasattr(self.pack, "set_row_blocks"):
                    self.pack.set_row_blocks(nc)      # (sweeps with one thread per (system, block))
        self.n_interior = len(rows)
        rows += flat(eq_sp)
        cols += flat(var_sp)
        if len(rows) != self.R or len(cols) != self.R:
            raise ValueError("gauge rows and columns do not balance")
        self.row_perm = np.array(rows, dtype=np.int32)
        self.col_perm = np.array(cols, dtype=np.int32)
        self.row_axes = np.zeros(self.R, dtype=np.uint8)
        self.col_axes = np.zeros(self.R, dtype=np.uint8)
        for i in self.eq_info:
            self.row_axes[i["row0"]:i["row0"] + i["rows"]] = i["bits"] | (0 if self.nf >= 2 else (2 if self.nf == 1 else 3))
        for i in self.var_info:
            self.col_axes[i["row0"]:i["row0"] + i["rows"]] = i["bits"] | (0 if self.nf >= 2 else (2 if self.nf == 1 else 3))
        rinv = np.empty(self.R, dtype=np.int64)
        cinv = np.empty(self.R, dtype=np.int64)
        rinv[self.row_perm] = np.arange(self.R)
        cinv[self.col_perm] = np.arange(self.R)
        kl = ku = 0
        for tl in (self.MP_tl, self.LP_tl):
            if tl.nterms == 0:
                continue
            i, c = rinv[tl.row], cinv[tl.col]
            m = (i < self.n_interior) & (c < self.n_interior) & (tl.dx == 0) & (tl.dy == 0)
            if m.any():
                kl = max(kl, int((i[m] - c[m]).max()))
                ku = max(ku, int((c[m] - i[m]).max()))
        nb = self.R - self.n_interior
        self.kl, self.ku = kl, max(ku, nb)
        self._build_grading()
        self._build_pairing()

    def _build_pairing(self):
        """x <-> y symmetry of the pencil matrices: lambda(ky, kx) = Pi_r lambda(kx, ky) Pi_c with Pi the involutions that
        swap the components of every tensor variable / equation along the two Fourier axes.  When it holds (equal box
        lengths and sizes, isotropic equations) the pack solves the pencil (my, mx) with the factorization of (mx, my)
        (ddh_pencil_set_pairing): half the factorizations, half the factor stream of every solve.  Verified here on the
        term lists of (a M + b L) P and of P; any mismatch leaves pairing off."""
        self.pairing = None
        if not hasattr(self.pack, "set_pairing") or os.environ.get("DDH_PAIR", "1") == "0":
            return
        if self.nf != 2 or self.real_grading is None or self.dist.size > 1:
            return
        nf, nx, ny, kx, ky = self.evaluator_core.geom()
        if nx != ny or not np.array_equal(np.asarray(kx), np.asarray(ky)):
            return
        sep = self.dist.separable_axes

        def swap_perm(infos, sig_of):
            perm = np.arange(self.R)
            for i in infos:
                sig = sig_of(i)
                dims = [cs.dim for cs in sig]
                if (int(np.prod(dims)) if dims else 1) != i["ncomp"]:
                    return None
                maps = []
                for cs in sig:
                    m = list(range(cs.dim))
                    idx = {self.dist.coord_axis(c): k for k, c in enumerate(cs.coords)}
                    if sep[0] in idx and sep[1] in idx:
                        m[idx[sep[0]]], m[idx[sep[1]]] = idx[sep[1]], idx[sep[0]]
                    elif sep[0] in idx or sep[1] in idx:
                        return None
                    maps.append(m)
                for comp in range(i["ncomp"]):
                    multi = np.unravel_index(comp, dims) if dims else ()
                    image = int(np.ravel_multi_index([maps[k][multi[k]] for k in range(len(dims))], dims)) if dims else 0
                    r0 = i["row0"] + comp * i["nz"]
                    perm[r0:r0 + i["nz"]] = i["row0"] + image * i["nz"] + np.arange(i["nz"])
            return perm

        try:
            col_swap = swap_perm(self.var_info, lambda i: i["field"].tensorsig)
            row_swap = swap_perm(self.eq_info, lambda i: i["eq"]["tensorsig"])
        except (AttributeError, KeyError, ValueError):
            return
        if col_swap is None or row_swap is None:
            return
        if np.array_equal(col_swap, np.arange(self.R)) and np.array_equal(row_swap, np.arange(self.R)):
            pass                                            # scalar problem: the swap is the identity, still a symmetry

        def symmetric(tl, rs, cs):
            if tl is None or tl.nterms == 0:
                return True
            a = tl.consolidated()
            from ..pencilpack import TermList
            b = TermList(a.nrows, a.ncols, rs[a.row], cs[a.col], a.coef, a.ey, a.ex, a.dy, a.dx).consolidated()
            if a.nterms != b.nterms:
                return False
            same = all(np.array_equal(getattr(a, k), getattr(b, k)) for k in ("row", "col", "ex", "ey", "dx", "dy"))
            scale = np.abs(a.coef).max()
            # real operators: coef i^-(ex+ey) real, so that lambda(-kx, -ky) = conj lambda(kx, ky)
            phase = a.coef * (-1j) ** ((a.ex.astype(int) + a.ey.astype(int)) % 4)
            return same and np.abs(a.coef - b.coef).max() <= 1e-12 * scale and np.abs(phase.imag).max() <= 1e-12 * scale

        if not (symmetric(self.MP_tl, row_swap, col_swap) and symmetric(self.LP_tl, row_swap, col_swap)
                and symmetric(self.P_tl, col_swap, col_swap)):
            return
        # the swap must respect the band / border split and the validity masks of the ordering
        if not (np.array_equal(self.row_axes[row_swap], self.row_axes) and np.array_equal(self.col_axes[col_swap], self.col_axes)):
            return
        self.pairing = dict(row_swap=row_swap.astype(np.int32), col_swap=col_swap.astype(np.int32))
        self.pack.set_pairing(self.pairing["row_swap"], self.pairing["col_swap"],
                              int(os.environ.get("DDH_PAIR_MIN", "65536")))

    def _build_grading(self):
        """Z2 gradings that make the pencil matrices real and shared by the +kx / -kx systems.

        For real differential operators every entry of the complex symbol carries the phase i^(order of
        differentiation); if rows and columns can be two-coloured so that phase(entry) = rot[row] +
        rot[col] (mod 2), then lambda = D_r A D_c^-1 with A real, D = diag(i^rot).  Likewise if
        parity(kx exponent) = sgn[row] + sgn[col], then lambda(-kx) = S_r lambda(kx) S_c.  Both are found by
        a breadth-first two-colouring of the bipartite row/column graph of (a M + b L) P; if either fails
        (e.g. dispersive KdV: real and imaginary terms in one entry) the general complex path is used."""
        from ..pencilpack import TermList
        self.real_grading = None
        if self.nf == 0 or self.n_interior == 0:
            return
        tls = [t for t in (self.MP_tl, self.LP_tl) if t.nterms]
        if not tls:
            return
        row = np.concatenate([t.row for t in tls])
        col = np.concatenate([t.col for t in tls])
        coef = np.concatenate([t.coef for t in tls])
        ex = np.concatenate([t.ex for t in tls])
        mag = np.abs(coef)
        is_re = np.abs(coef.imag) <= 1e-13 * mag
        is_im = np.abs(coef.real) <= 1e-13 * mag
        if not np.all(is_re | is_im):
            return
        rot = _two_colour(self.R, row, col, np.where(is_re, 0, 1))
        if rot is None:
            return
        if self.nf == 2:
            sgn = _two_colour(self.R, row, col, (ex % 2).astype(int))
            if sgn is None:
                return
        else:
            sgn = (np.zeros(self.R, dtype=int), np.zeros(self.R, dtype=int))
        rr, rc = rot
        sr, sc = sgn

        def graded(tl):
            c = tl.coef * (1j) ** ((rc[tl.col] - rr[tl.row]) % 4)
            assert np.all(np.abs(c.imag) <= 1e-12 * np.maximum(np.abs(c), 1e-300)), "grading did not make the matrix real"
            return TermList(tl.nrows, tl.ncols, tl.row, tl.col, c.real.astype(complex), tl.ex, tl.ey, tl.dx, tl.dy)

        self.real_grading = dict(matM=self.pack.add_matrix(graded(self.MP_tl)),
                                 matL=self.pack.add_matrix(graded(self.LP_tl)),
                                 row_code=(rr | (sr << 1)).astype(np.uint8),
                                 col_code=(rc | (sc << 1)).astype(np.uint8))

    # ---- RHS plan --------------------------------------------------------------------------------------------
    def _build_F_plan(self):
        ev = self.evaluator_core
        ctx = LinCtx(self.variables, strict=False)
        nl_leaves, nl_rows = [], 0
        groups = {"nl": [], "x": [], "const": [], "param": {}}
        for einfo in self.eq_info:
            F = einfo["eq"]["F"]
            if F is None:
                continue
            le = F.lin(ctx)
            efx, efy = self._flags(einfo["eq"]["domain"])
            for leaf, terms in le.leaves.items():
                if not terms:
                    continue
                lfx, lfy = self._flags(leaf.domain)
                nzi = self.dist.coupled_size(leaf.domain)
                blk = [einfo["row0"], le.nzo, None, nzi, terms, efx or lfx, efy or lfy]
                if isinstance(leaf, Field):
                    if getattr(leaf, "_sysbuf", None) is self.sysbuf:
                        blk[2] = leaf._row0
                        groups["x"].append(tuple(blk))
                    elif getattr(leaf, "_is_number", False):
                        groups["const"].append((leaf, blk))
                    else:
                        groups["param"].setdefault(id(leaf), [leaf, []])[1].append(blk)
                else:
                    if not _full_sep(self.dist, leaf.domain):
                        raise NotImplementedError("nonlinear RHS term without all Fourier bases")
                    found = [x for x in nl_leaves if x[0] is leaf]
                    if found:
                        row0 = found[0][1]
                    else:
                        row0 = nl_rows
                        nl_leaves.append((leaf, row0, leaf.ncomp * nzi))
                        nl_rows += leaf.ncomp * nzi
                    blk[2] = row0
                    groups["nl"].append(tuple(blk))
        self.nl_leaves, self.nl_rows = nl_leaves, nl_rows
        # fused grid stage: product leaves that share their register operand run in one launch
        self.nl_fused, self.nl_plain = [], []
        lim = self.ex.FUSED_LIMITS
        for item in nl_leaves:
            fp = ev.fusable_product(item[0]) if os.environ.get("DDH_NO_FUSED_GRID") is None else None
            if fp is None:
                self.nl_plain.append(item)
                continue
            for grp in self.nl_fused:
                g0 = grp[0][1]
                nbc = sum({id(x[1][1]): x[1][1].ncomp for x in grp + [(None, fp)]}.values())
                if (g0[0] is fp[0] and g0[4] == fp[4] and nbc <= lim["nb"]
                        and sum(x[0][0].ncomp for x in grp) + item[0].ncomp <= lim["nc"]
                        and sum(len(x[1][2]) for x in grp) + len(fp[2]) <= lim["terms"]):
                    grp.append((item, fp))
                    break
            else:
                self.nl_fused.append([(item, fp)])
        nf, nx, ny, kx, ky = ev.geom()
        self.F_nl = None
        self._plan_direct_F(groups)
        if nl_rows:
            self.NLbuf = None if self.F_direct is not None else self.ex.zeros((nl_rows, nx, ny))
            self.nl_pack = self.ex.make_pack(nf, nl_rows, nx, ny, kx, ky, self.dist._mx_offset)
            self.F_nl = self.nl_pack.add_matrix(self._eq_transform(flatten(groups["nl"], self.R, nl_rows)))
        self.F_x = None
        if groups["x"]:
            self.F_x = self.pack.add_matrix(self._eq_transform(flatten(groups["x"], self.R, self.R)))
        self.F_params = []
        for key, (leaf, blks) in groups["param"].items():
            rows = leaf.ncomp * self.dist.coupled_size(leaf.domain)
            pk = self.ex.make_pack(nf, rows, nx, ny, kx, ky, self.dist._mx_offset)
            mid = pk.add_matrix(self._eq_transform(flatten([tuple(b[:2] + [0] + b[3:]) for b in blks], self.R, rows)))
            self.F_params.append((leaf, pk, mid))
        # constants (e.g. "b(z=0) = Lz"): evaluated once
        self.F_const = None
        if groups["const"]:
            total = np.zeros((self.R, nx, ny))
            for leaf, blk in groups["const"]:
                tl = flatten([tuple(blk[:2] + [0] + blk[3:])], self.R, 1)
                val = leaf._number
                for t in range(tl.nterms):
                    # constants only reach the k=0 pencil, cos-cos part
                    if tl.ex[t] == 0 and tl.ey[t] == 0 and self.dist._mx_offset == 0:
                        total[tl.row[t], 0, 0] += (tl.coef[t] * val).real
            if self.eq_T is not None:                    # (boundary rows: the equivalent combinations, eq_T)
                total = (self.eq_T @ total.reshape(self.R, -1)).reshape(total.shape)
            nz = np.flatnonzero(total)
            self._F_const_index = nz
            self.F_const = self.ex.make_scatter(nz, total.reshape(-1)[nz]) if nz.size else None
            self._F_const_rows = np.unique(nz // (nx * ny))

    def _plan_direct_F(self, groups):
        """Right-hand sides that are nothing but fused nonlinear products converted to their equation's basis (times a
        constant): F_u = -u.grad(u) and the like.  The forward Jacobi-axis transform then writes the equation's rows of the
        F system vector itself -- the conversion T -> (a, b) of the equation is the transform's own band apply
        (core/transforms.py:862-874), the constant goes into the product's term coefficients -- and the gather mat-vec
        (nl_pack / F_nl: read 4, write R row blocks, per stage) never runs.  self.F_direct = [(group index, member index,
        equation info, scale)] or None when any part of F needs the general path."""
        self.F_direct = None
        if os.environ.get("DDH_NO_DIRECT_F") is not None or not self.nl_leaves or self.nl_plain:
            return
        if groups["x"] or groups["param"]:
            return
        from ..tools import jacobi
        by_row0 = {}
        for blk in groups["nl"]:
            by_row0.setdefault(blk[2], []).append(blk)
        plan, eq_rows = {}, []
        for leaf, row0, rows in self.nl_leaves:
            blks = by_row0.get(row0, [])
            if len(blks) != 1:
                return
            eq_row0, nzo, _, nzi, terms, fx, fy = blks[0]
            einfo = [e for e in self.eq_info if e["row0"] == eq_row0]
            if not einfo or fx or fy or nzo != nzi or einfo[0]["ncomp"] != leaf.ncomp or len(terms) != leaf.ncomp:
                return
            einfo = einfo[0]
            jac = [ax for ax in self.dist._jacobi_axes if leaf.domain.by_axis[ax] is not None]
            if len(jac) != 1:
                return
            bl, be = leaf.domain.by_axis[jac[0]], einfo["eq"]["domain"].by_axis[jac[0]]
            if be is None or (bl.a, bl.b) != (bl.a0, bl.b0) or (be.a0, be.b0) != (bl.a0, bl.b0) or (be.a, be.b) == (bl.a, bl.b):
                return
            for ax, b in enumerate(leaf.domain.by_axis):
                if ax != jac[0] and b is not einfo["eq"]["domain"].by_axis[ax]:
                    return
            conv = jacobi.conversion_matrix(nzi, bl.a0, bl.b0, be.a, be.b).tocsr()
            offs = np.unique((conv.tocoo().col - conv.tocoo().row))
            if offs.min() < 0 or len(offs) > 4:
                return
            scale = None
            for c, t in enumerate(sorted(terms, key=lambda t: t.co)):
                if t.co != c or t.ci != c or t.ex or t.ey or t.dx or t.dy or t.dt or t.coef.imag != 0.0:
                    return
                Z = t.Z.tocsr() * t.coef.real
                if Z.shape != conv.shape:
                    return
                ratio = Z.diagonal()[0] / conv.diagonal()[0]
                if scale is None:
                    scale = ratio
                diff = abs(Z - scale * conv)
                if diff.nnz and diff.max() > 1e-13 * abs(scale) * abs(conv).max():
                    return
            plan[id(leaf)] = (einfo, float(scale))
            eq_rows.append((eq_row0, eq_row0 + einfo["rows"]))
        # constants of F must not fall into rows the transforms overwrite
        for leaf, blk in groups["const"]:
            lo, hi = blk[0], blk[0] + blk[1]
            if any(lo < b and a < hi for a, b in eq_rows):
                return
        self.F_direct = plan
        self._F_zeroed = set()

    def evaluate_F(self, out, persistent=False, tiled_row=0):
        """F system vector for the current state (coefficient space, equation bases).
        persistent: `out` is a buffer the caller owns for the life of the solver and nothing else writes to (the
        timesteppers' F arrays): rows without right-hand-side terms are then zeroed once.  Any other buffer -- a temporary
        whose address the allocator may hand out again after unrelated data lived there -- is zeroed on every call."""
        ev, ex = self.evaluator_core, self.ex
        ev.new_pass()
        work = self.__dict__.pop("_overlap_work", None) or []    # (state-only work of the caller: overlaps_rhs_work)
        if self._grid_windows() == 1:
            for f in work:
                f()
            work = []
            ev.prefetch_stage1()                # several ranks (DDH_A2A_PREFETCH): every field's z step + exchange start first
        if tiled_row and self.F_direct is None:
            raise RuntimeError("tile-major right-hand sides need the direct-F plan")
        if self.F_direct is not None:
            tr = self.dist.transformer
            key = (out.data_ptr() if hasattr(out, "data_ptr") else id(out))
            if not persistent or key not in self._F_zeroed:     # rows without right-hand-side terms are zero and stay zero
                ex.fill_zero(out)
                if persistent:
                    self._F_zeroed.add(key)
            K = self._grid_windows()
            if K > 1:
                self._evaluate_F_windows(out, K, tiled_row, work)
                ev.new_pass()
                if self.F_const is not None:
                    ex.scatter_set(out, self.F_const)
                return
            for grp in self.nl_fused:
                outs = [ex.empty(tr.pregrid_shape(leaf.domain, leaf.ncomp, leaf.domain.dealias))
                        for (leaf, row0, rows), fp in grp]
                ev.eval_fused_products([(item[0], fp) for item, fp in grp], outs,
                                       scales=[self.F_direct[id(item[0])][1] for item, fp in grp])
                finish = []
                for ((leaf, row0, rows), fp), pg in zip(grp, outs):
                    einfo = self.F_direct[id(leaf)][0]
                    edom = einfo["eq"]["domain"]
                    dst = out[einfo["row0"]:einfo["row0"] + einfo["rows"]].reshape(
                        (leaf.ncomp,) + tuple(edom.storage_coeff_shape()))
                    # every product's transforms up to its pencil transpose first, then the rest: on several ranks the
                    # exchange of one product runs under the x transforms of the next (Transformer.forward_begin)
                    finish.append(tr.forward_begin(edom, leaf.ncomp, pg, edom.dealias, dst, skip_last=True, tiled_row=tiled_row))
                for f in finish:
                    f()
            ev.new_pass()
            if self.F_const is not None:
                ex.scatter_set(out, self.F_const)       # (entries of the kx = ky = 0 cos-cos mode: offset 0 of a row in
            return                                       #  the natural and in the tile-major layout alike)
        parts = []

        def target():
            """The first full-size contribution is written straight into `out`."""
            y = out if not parts else ex.empty((self.R, self.nx, self.ny))
            parts.append(y)
            return y

        if self.F_nl is not None:
            tr = self.dist.transformer
            for grp in self.nl_fused:
                outs = [ex.empty(tr.pregrid_shape(leaf.domain, leaf.ncomp, leaf.domain.dealias))
                        for (leaf, row0, rows), fp in grp]
                ev.eval_fused_products([(item[0], fp) for item, fp in grp], outs)
                for ((leaf, row0, rows), fp), pg in zip(grp, outs):
                    dst = self.NLbuf[row0:row0 + rows].reshape((leaf.ncomp,) + tuple(leaf.domain.storage_coeff_shape()))
                    tr.forward_data(leaf.domain, leaf.ncomp, pg, leaf.domain.dealias, dst, skip_last=True)
            for leaf, row0, rows in self.nl_plain:
                g = ev.eval_grid(leaf)
                dst = self.NLbuf[row0:row0 + rows].reshape((leaf.ncomp,) + tuple(leaf.domain.storage_coeff_shape()))
                tr.forward_data(leaf.domain, leaf.ncomp, g, leaf.domain.dealias, dst)
            self.nl_pack.matvec(self.F_nl, self.NLbuf, target())
        if self.F_x is not None:
            self.pack.matvec(self.F_x, self.X, target())
        for leaf, pk, mid in self.F_params:
            pk.matvec(mid, ev._leaf_plane_data(leaf), target())
        ev.new_pass()
        if not parts:
            ex.fill_zero(out)
        elif len(parts) > 1:
            ex.lincomb(out, parts, [1.0] * len(parts))      # parts[0] is out itself
        if self.F_const is not None:
            ex.scatter_add(out, self.F_const)               # a handful of entries of the k = 0 pencil

    def _grid_windows(self):
        """Number of windows of z planes the grid stage of a sharded evaluation runs in (Transformer windows): several
        ranks with the blocked stage layout, every product through the fused stage, every operand through the stage
        cache (known after the first, un-windowed pass), DDH_A2A_WINDOWS = K dividing this rank's planes (default 2: under
        an emulated 75 GB/s wire a P = 8 rank steps in 10.0 instead of 12.6 ms, 4 windows in 10.4; 1 = off)."""
        K = int(os.environ.get("DDH_A2A_WINDOWS", "2"))
        if K <= 1 or self.dist.size == 1 or self.F_direct is None or not self.nl_fused:
            return 1
        ev, tr = self.evaluator_core, self.dist.transformer
        if not ev.windows_possible() or not hasattr(tr, "windows_begin") or os.environ.get("DDH_A2A_DEFER", "1") == "0":
            return 1
        for grp in self.nl_fused:
            for (leaf, row0, rows), fp in grp:
                edom = self.F_direct[id(leaf)][0]["eq"]["domain"]
                if tr.stage_xb(edom, edom.dealias) is None or tr.stage_xb(leaf.domain, leaf.domain.dealias) is None:
                    return 1
                gz = tr.pregrid_shape(leaf.domain, leaf.ncomp, leaf.domain.dealias, window=False)[1]
                if gz % K:
                    return 1
        return K

    def overlaps_rhs_work(self):
        """True when evaluate_F has a wait for the wire in it (the grid stage in windows on several ranks): the caller may
        hand it state-only work (`_overlap_work`: the stage's mat-vecs) to issue under that wait instead of in front of it."""
        return self._grid_windows() > 1

    def _evaluate_F_windows(self, out, K, tiled_row, work=()):
        """evaluate_F (direct branch) with the grid stage in K windows of z planes: z steps of every operand field, their
        exchanges queued window by window, then per window the x steps, the fused launches and the x forward steps -- whose
        parts of the forward exchange are on the wire while the next window computes -- and at the end the z forward steps."""
        ev, ex, tr = self.evaluator_core, self.ex, self.dist.transformer
        tr.windows_begin(K)
        try:
            ev.prefetch_stage1(force=True)              # every field's z step; the transposes are registered, not started
            tr.windows_start_exchanges()
            fwd = {}
            for k in range(K):
                tr.window_set(k)
                ev.window_cache_reset()
                for gi, grp in enumerate(self.nl_fused):
                    outs = [ex.empty(tr.pregrid_shape(leaf.domain, leaf.ncomp, leaf.domain.dealias))
                            for (leaf, row0, rows), fp in grp]
                    ev.eval_fused_products([(item[0], fp) for item, fp in grp], outs,
                                           scales=[self.F_direct[id(item[0])][1] for item, fp in grp])
                    for pi, (((leaf, row0, rows), fp), pg) in enumerate(zip(grp, outs)):
                        if (gi, pi) not in fwd:
                            einfo = self.F_direct[id(leaf)][0]
                            edom = einfo["eq"]["domain"]
                            dst = out[einfo["row0"]:einfo["row0"] + einfo["rows"]].reshape(
                                (leaf.ncomp,) + tuple(edom.storage_coeff_shape()))
                            tr.window_set(None)
                            fwd[(gi, pi)] = tr.forward_windows(edom, leaf.ncomp, edom.dealias, dst, tiled_row=tiled_row)
                            tr.window_set(k)
                        fwd[(gi, pi)].push(k, pg)
            tr.window_set(None)
            for f in work:                              # the caller's state-only kernels: under the last window's departure
                f()
            for f in fwd.values():
                f.finish()
        finally:
            tr.windows_end()

    # ---- un-aliased variables (no separable bases): tiny copies around each solve ---------------------------
    def push_unaliased(self):
        for info in self.var_info:
            if not info["aliased"]:
                v = info["field"]
                c = v.coeff_data()
                self._plane_copy(info, v, c, to_state=True)

    def pull_unaliased(self):
        for info in self.var_info:
            if not info["aliased"]:
                v = info["field"]
                self._plane_copy(info, v, v._alloc_c(), to_state=False)
                v.mark_device_coeff_current()

    def _plane_copy(self, info, v, c, to_state):
        sep = self.dist.separable_axes
        sx = 1 if (self.nf < 1 or v.domain.by_axis[sep[0]] is None) else self.nx
        sy = 1 if (self.nf < 2 or v.domain.by_axis[sep[1]] is None) else self.ny
        if sx == 1 and self.nf >= 1 and self.dist._mx_offset != 0:
            return          # data without x dependence belongs to the kx = 0 pencils of the first rank
        if getattr(self, "x_banded", 0):            # (sx = sy = 1, _state_tiling_ok: entry (row, 0, 0) of a kx-band-major state)
            xs = self.X.as_strided((info["rows"], 1, 1), (8 * self.ny, 1, 1), self.X.storage_offset() + info["row0"] * 8 * self.ny)
        else:
            xs = self.X[info["row0"]:info["row0"] + info["rows"]].reshape(info["rows"], self.nx, self.ny)[:, :sx, :sy]
        cs = c.reshape(info["rows"], sx, sy)
        if to_state:
            self.ex.assign(xs, cs)
        else:
            self.ex.assign(cs, xs)

    def mark_state_current(self):
        self.sysbuf.invalidate()                # (a tile-major state: the natural shadows of its rows are stale)
        for info in self.var_info:
            if info["aliased"]:
                info["field"].mark_device_coeff_current()
        self.pull_unaliased()

    def sync_state_to_device(self):
        for info in self.var_info:
            info["field"].coeff_tiled()         # (host edits uploaded; a tile-major field is NOT converted for this)
        self.push_unaliased()

    # ---- tile-major state vector ---------------------------------------------------------------------------
    def _state_tiling_ok(self):
        ex = self.ex
        if os.environ.get("DDH_X_TILED", "1") == "0" or not hasattr(ex, "tile_rows") or not hasattr(self.pack, "set_state_tiled"):
            return False
        if self.nf != 2 or self.nx % 8 or self.ny % 8 or self.dist.size > 1:
            return False
        if self.nx * self.ny < int(os.environ.get("DDH_X_TILED_MIN", 4 * 16384)):
            return False
        sep = self.dist.separable_axes
        tr = self.dist.transformer
        for info in self.var_info:
            v = info["field"]
            if info["aliased"]:
                if tuple(v._storage_shape("c", None))[-2:] != (self.nx, self.ny):
                    return False
                # a field with a Jacobi axis is an operand of the right-hand side: its backward z transform must read the
                # tile-major rows in place (the strided wave kernels' sizes), or every evaluation would convert it first
                has_jacobi = any(v.domain.by_axis[ax] is not None for ax in self.dist._jacobi_axes)
                if has_jacobi and not tr.coeff_tiled_ok(v.domain, v.domain.dealias, self.ny):
                    return False
            elif any(v.domain.by_axis[ax] is not None for ax in sep):
                return False                    # (plane copies of a field with one Fourier axis address X naturally)
        return True

    def _enable_state_tiling(self):
        """Keep the state vector X tile-major ([kx / 8][ky / 8][kx % 8][ky % 8] within a row, like the right-hand-side
        vectors): a wavefront of the backward sweep stores a solution row as two 512-byte runs instead of sixteen 64-byte
        runs over eight storage rows (solve -0.5 ms per launch pair at 512^2 pencils).  The kernels of the step follow --
        every solve of the pack writes, every mat-vec reads that layout (ddh_pencil_set_state_tiled), the backward z
        transforms read it (ddh_fft_set_coeff_tiled; Evaluator._stage0) -- and everything else sees natural shadows of the
        rows it asks for (SystemBuffer, Field.require_coeff_space).  One rank, two Fourier axes with sizes that are
        multiples of 8; DDH_X_TILED=0 keeps the natural layout.  Replaces nothing in the reference, whose state lives in
        the fields (core/subsystems.py:497-596 gathers / scatters per solve)."""
        self.x_tiled = 0
        if not self._state_tiling_ok():
            return
        ex = self.ex
        # DDH_X_TILED=2: kx-band-major, the rows of a band of 8 storage rows together (32 KiB instead of 2 MiB between the rows
        # of a pencil).  Measured at 512^2 x 256: the backward z transform 1.20 -> 1.16 ms, everything else unchanged -- the
        # 2 MiB stride is not what limits the sweeps -- and the unfused recombination (small problems) has no such form: opt-in.
        mode = 2 if os.environ.get("DDH_X_TILED", "1") == "2" else 1
        self.sync_state_to_device()
        tmp = ex.empty((self.R, self.nx, self.ny))
        ex.tile_rows(self.X, tmp, self.R, self.nx, self.ny, True, self.R if mode == 2 else 0)
        ex.copy(self.X, tmp)
        del tmp
        sb = self.sysbuf
        sb.tiled = int(self.ny)
        sb.banded = int(self.R) if mode == 2 else 0
        for info in self.var_info:
            sb.ranges.append((info["row0"], info["rows"]))
            sb.valid[info["row0"]] = False
            if info["aliased"]:
                v = info["field"]
                v._tiled, v._nrows = sb, info["rows"]
                v._c = sb.state_rows(info["row0"], info["rows"])
        self.pack.set_state_tiled(mode)
        self.x_tiled = int(self.ny)
        self.x_banded = sb.banded

    def state_natural(self):
        """The state vector [R][nx][ny] in the natural layout (tests, tools): X itself, or the natural shadow of a
        tile-major state brought up to date."""
        return self.sysbuf.natural(self.ex)

    def rhs_tiling(self, lus):
        """Row length ny when the solver-internal right-hand-side vectors (the timestepper's M.X and F buffers) can use
        the TILE-MAJOR layout [kx/8][ky/8][kx%8][ky%8] the sweeps read contiguously (ddh_pencil_solve_recombined_tiled,
        include/dedalus_hip.h), else 0.  Needs: two Fourier axes with sizes that are multiples of 8, right-hand sides
        written by the forward transforms themselves (direct F) through the strided-axis wave kernel, the window-form M.X
        product -- properties of the problem, decided once -- and the lean forward sweep over real factors for EVERY
        factorization in `lus` (an id or an iterable of ids), asked again on every call: the sweep variant belongs to a
        factorization and to PencilPack.set_solve_variant, not to the solver.  DDH_NO_RHS_TILING=1 switches it off (A/B)."""
        if getattr(self, "_rhs_tiling_problem", None) is None:
            self._rhs_tiling_problem = self._rhs_tiling_of_problem()
        if not self._rhs_tiling_problem:
            return 0
        for lu in ([lus] if isinstance(lus, (int, np.integer)) else list(lus)):
            info = self.pack.lu_info(lu)
            if info["forward"] != "lean" or not info["real"]:
                return 0
        return self._rhs_tiling_problem

    def _rhs_tiling_of_problem(self):
        ex = self.ex
        if (os.environ.get("DDH_NO_RHS_TILING") is not None or not getattr(self.pack, "supports_tiled_rhs", False)
                or self.nf != 2 or self.nx % 8 or self.ny % 8 or self.F_direct is None or self.P_id is None
                or self.real_grading is None or self.nx * self.ny < int(os.environ.get("DDH_RHS_TILING_MIN", 4 * 16384))
                or not hasattr(ex, "tiled_forward_ok")):
            return 0
        if self.F_const is not None and np.any(np.asarray(self._F_const_index) % (self.nx * self.ny) != 0):
            return 0
        tr = self.dist.transformer
        for einfo, _scale in self.F_direct.values():
            edom = einfo["eq"]["domain"]
            pos, b, spec = tr._steps(edom, edom.dealias)[0]
            if pos != 0 or not ex.tiled_forward_ok(spec, b, self.nx * self.ny, self.ny):
                return 0
        return int(self.ny)

    def untile_rows(self, vec, banded=False):
        """Natural-layout copy [R][nx][ny] of a tile-major system vector (diagnostics / parity probes only); banded: of a
        kx-band-major solution vector."""
        R, nx, ny = self.R, self.nx, self.ny
        if banded:                                      # [kx / 8][R][ky / 8][8][8]
            return vec.reshape(nx // 8, R, ny // 8, 8, 8).permute(1, 0, 3, 2, 4).contiguous().reshape(R, nx, ny)
        v = vec.reshape(R, nx // 8, ny // 8, 8, 8)
        if hasattr(v, "permute"):
            return v.permute(0, 1, 3, 2, 4).contiguous().reshape(R, nx, ny)
        return np.ascontiguousarray(np.transpose(v, (0, 1, 3, 2, 4))).reshape(R, nx, ny)

    def solve(self, lu, rhs, out):
        """out = (a M + b L)^-1 rhs  through the recombined band factorization: X = P Y."""
        if self.P_id is None:
            self.pack.solve(lu, rhs, out)
        elif hasattr(self.pack, "solve_recombined"):
            Y = self.ex.empty((self.R, self.nx, self.ny))
            self.pack.solve_recombined(lu, [rhs], [1.0], self.P_id, Y, out)
        else:
            Y = self.ex.empty((self.R, self.nx, self.ny))
            self.pack.solve(lu, rhs, Y)
            self.pack.matvec(self.P_id, Y, out)
        if getattr(self, "solve_probe", None) is not None:
            self._probe_record(lu, rhs, out, path="ddh_pencil_solve_recombined (one materialised right-hand side)")

    def _probe_record(self, lu, rhs, out, path, skip_rows=None, terms=1, zero_rows=None):
        """Parity probe (tests, bench.py `parity`): the right-hand side and the solution of the solve that just ran, on the
        probe's pencils, in the reference's gathered order.  `skip_rows` (host uint8 [R] or None): unknowns this solve did
        not store -- their entries of `x` are stale by design and the record says which."""
        probe = self.solve_probe
        a, b = self._lu_params[lu]
        if getattr(self, "x_tiled", 0):
            out = self.untile_rows(out, banded=bool(getattr(self, "x_banded", 0)))  # (every solution vector of such a pack)
        rec = dict(a=a, b=b, path=path, terms=int(terms), zero_rows=bool(zero_rows), skip_rows=skip_rows is not None,
                   rhs=[self.gather_pencil(rhs, "equations", gx, gy) for gx, gy in probe["groups"]],
                   x=[self.gather_pencil(out, "variables", gx, gy) for gx, gy in probe["groups"]])
        if skip_rows is not None:
            rec["skipped"] = [self.gather_rows_mask(skip_rows, "variables", gx, gy) for gx, gy in probe["groups"]]
        probe["records"].append(rec)

    def zero_rows_host(self):
        """uint8 [R]: 1 where BOTH M.X and the F vector of `evaluate_F` are structurally zero -- rows of equations without a
        time derivative whose right-hand side is neither written by the direct-F transforms nor a constant (the continuity
        equation, homogeneous boundary rows); None when F goes through the general gather path (its rows are not known
        here)."""
        if self.F_direct is None:
            return None
        mask = np.ones(self.R, dtype=np.uint8)
        mask[np.unique(np.asarray(self.M_tl.row))] = 0
        for einfo, _scale in self.F_direct.values():
            mask[einfo["row0"]:einfo["row0"] + einfo["rows"]] = 0
        if self.F_const is not None:
            mask[self._F_const_rows] = 0
        return mask

    def skip_rows_host(self):
        """uint8 [R]: 1 for the unknowns that nothing reads between the stages of a Runge-Kutta step -- variables without a
        column in the mass matrix M that are not operands of any F expression (pressure, tau variables)."""
        used = set()
        for einfo in self.eq_info:
            F = einfo["eq"]["F"]
            if F is not None and hasattr(F, "leaves"):
                used |= {id(f) for f in F.leaves()}
        mask = np.zeros(self.R, dtype=np.uint8)
        mcols = np.zeros(self.R, dtype=bool)
        mcols[np.unique(np.asarray(self.M_tl.col))] = True
        for info in self.var_info:
            r0, r1 = info["row0"], info["row0"] + info["rows"]
            if id(info["field"]) not in used and not mcols[r0:r1].any():
                mask[r0:r1] = 1
        return mask

    def _device_mask(self, mask):
        t = self.ex.torch
        return (t.as_tensor(mask, device=self.ex.dev.tdev), float(mask.mean()))

    def mx_f_zero_rows(self):
        """Device form (byte mask, fraction set) of `zero_rows_host`, or None.  A Runge-Kutta right-hand side built from M.X
        and F vectors only (timesteppers.RungeKuttaIMEX) passes it to the solve, whose forward sweep then does not read
        those rows (ddh_pencil_solve_recombined_sparse)."""
        if getattr(self, "_zrows", "unset") != "unset":
            return self._zrows
        self._zrows = None
        if not getattr(self.pack, "supports_zero_rows", False) or self.P_id is None:
            return None
        if os.environ.get("DDH_NO_ZERO_ROWS") is not None:
            return None
        mask = self.zero_rows_host()
        if mask is not None and mask.any():
            self._zrows = self._device_mask(mask)
        return self._zrows

    def intermediate_skip_rows(self):
        """Device form of `skip_rows_host`, or None: the solve of an INTERMEDIATE Runge-Kutta stage does not store these
        unknowns (they are final only after the last stage)."""
        if getattr(self, "_skiprows", "unset") != "unset":
            return self._skiprows
        self._skiprows = None
        if not getattr(self.pack, "supports_zero_rows", False) or self.P_id is None or self.F_direct is None:
            return None
        if os.environ.get("DDH_NO_SKIP_ROWS") is not None:
            return None
        mask = self.skip_rows_host()
        if mask.any():
            self._skiprows = self._device_mask(mask)
        return self._skiprows

    def solve_lincomb(self, lu, xs, alphas, out, zero_rows=None, skip_rows=None, tiled=False):
        """out = (a M + b L)^-1 (sum_t alphas[t] xs[t]).  The combination is formed inside the forward sweep of the band
        solve; with a parity probe attached (or more terms than the kernel takes) it is materialised first."""
        probe = getattr(self, "solve_probe", None)
        if tiled and (self.P_id is None or not hasattr(self.pack, "solve_recombined") or len(xs) > self.pack.MAX_RHS_TERMS):
            raise RuntimeError("tile-major right-hand sides need the fused recombined solve")
        if len(xs) > self.pack.MAX_RHS_TERMS or not hasattr(self.pack, "solve_lincomb"):
            rhs = self.ex.empty((self.R, self.nx, self.ny))
            self.ex.lincomb(rhs, xs, alphas)
            self._last_rhs = rhs                       # (kept for the parity tests)
            return self.solve(lu, rhs, out)
        rhs_rec = None
        if probe is not None:
            # The probe must certify the kernels that are TIMED: the same fused call runs below, with the same terms and
            # masks; the record's right-hand side is formed independently (a plain lincomb of the same vectors, every row),
            # so a mask that hid a non-zero row would show up as a residual.  (`out` may alias a term: formed first.)
            rhs_rec = self.ex.empty((self.R, self.nx, self.ny))
            self.ex.lincomb(rhs_rec, xs, alphas)
            if tiled:
                rhs_rec = self.untile_rows(rhs_rec)          # (the record and the tests' re-solves use the natural layout)
            self._last_rhs = rhs_rec
        path = "ddh_pencil_solve_lincomb"
        if self.P_id is None:
            self.pack.solve_lincomb(lu, xs, alphas, out)
        elif hasattr(self.pack, "solve_recombined"):
            Y = self.ex.empty((self.R, self.nx, self.ny))
            if tiled:
                self.pack.solve_recombined(lu, xs, alphas, self.P_id, Y, out, zero_rows=zero_rows, skip_rows=skip_rows,
                                           tiled=True)
                path = "ddh_pencil_solve_recombined_tiled"
            elif zero_rows is not None or skip_rows is not None:
                self.pack.solve_recombined(lu, xs, alphas, self.P_id, Y, out, zero_rows=zero_rows, skip_rows=skip_rows)
                path = "ddh_pencil_solve_recombined_sparse"
            else:
                self.pack.solve_recombined(lu, xs, alphas, self.P_id, Y, out)
                path = "ddh_pencil_solve_recombined"
        else:
            Y = self.ex.empty((self.R, self.nx, self.ny))
            self.pack.solve_lincomb(lu, xs, alphas, Y)
            self.pack.matvec(self.P_id, Y, out)
        if probe is not None:
            skip_h = None
            if skip_rows is not None:
                skip_h = np.asarray(skip_rows[0].cpu().numpy() if hasattr(skip_rows[0], "cpu") else skip_rows[0], dtype=np.uint8)
            self._probe_record(lu, rhs_rec, out, path + " (%d fused terms%s%s)" % (
                len(xs), ", zero_rows" if zero_rows is not None else "", ", skip_rows" if skip_rows is not None else ""),
                skip_rows=skip_h, terms=len(xs), zero_rows=zero_rows is not None)
        if skip_rows is not None and os.environ.get("DDH_POISON_SKIPPED") is not None:
            # debug aid: the unknowns an intermediate stage does not store must not be consumed before the last stage
            # stores them -- poison them and a consumer shows up as NaN in the end state (tests/test_gpu_ivp.py)
            t = self.ex.torch
            idx = t.nonzero(skip_rows[0]).flatten()
            if getattr(self, "x_banded", 0):
                out.reshape(self.nx // 8, self.R, -1)[:, idx] = float("nan")
            else:
                out.reshape(self.R, -1)[idx] = float("nan")

    def gather_pencil(self, vec, which, gx, gy=0):
        """One pencil of a system vector in the reference's gathered order (Subproblem.gather_inputs /
        gather_outputs before pre_right_pinv / pre_left, core/subsystems.py:302-365): for every variable (or equation)
        in problem order its slice [component..., 2 modes of x group gx, 2 modes of y group gy, all z], C order;
        operands without a Fourier basis belong to group 0 only.  Diagnostics / parity checks: a few KB per call."""
        lx = gx - (self.dist._mx_offset if self.nf >= 1 else 0)
        # the cell's slab [R][<= 2][<= 2] of the vector comes to the host in one copy
        cx = slice(2 * lx, 2 * lx + 2) if self.nf >= 1 else slice(0, 1)
        cy = slice(2 * gy, 2 * gy + 2) if self.nf >= 2 else slice(0, 1)
        slab = vec.reshape(self.R, self.nx, self.ny)[:, cx, cy]
        cell = np.array(self.ex.download(slab.contiguous() if hasattr(slab, "contiguous") else slab), dtype=float)
        cell = cell.reshape(self.R, cell.shape[-2], cell.shape[-1])
        if which != "variables" and self.eq_Tinv is not None:
            # boundary rows are stored as the combinations eq_T of the user's equations: hand out the user's
            rows = self._eq_T_rows
            cell[rows] = (self.eq_Tinv[rows][:, rows] @ cell[rows].reshape(len(rows), -1)).reshape(cell[rows].shape)
        return self._gather_cell(cell, which, gx, gy)

    def gather_rows_mask(self, mask, which, gx, gy=0):
        """A per-row flag [R] in the gathered order of `gather_pencil` (one entry per gathered mode)."""
        wx = 2 if self.nf >= 1 else 1
        wy = 2 if self.nf >= 2 else 1
        cell = np.broadcast_to(np.asarray(mask, dtype=float)[:, None, None], (self.R, wx, wy))
        return self._gather_cell(cell, which, gx, gy) != 0

    def _gather_cell(self, cell, which, gx, gy):
        infos = self.var_info if which == "variables" else self.eq_info
        parts = []
        for info in infos:
            bits = info["bits"]
            if self.nf >= 1:
                sx = slice(0, 2) if (bits & 1) else (slice(0, 1) if gx == 0 else slice(0, 0))
            else:
                sx = slice(0, 1)
            if self.nf >= 2:
                sy = slice(0, 2) if (bits & 2) else (slice(0, 1) if gy == 0 else slice(0, 0))
            else:
                sy = slice(0, 1)
            blk = cell[info["row0"]:info["row0"] + info["rows"]].reshape(info["ncomp"], info["nz"], cell.shape[1], cell.shape[2])
            blk = blk[:, :, sx, sy].transpose(0, 2, 3, 1)
            if self.nf < 2:
                blk = blk[:, :, 0, :] if self.nf == 1 else blk[:, 0, 0, :]
            parts.append(np.ascontiguousarray(blk).ravel())
        return np.concatenate(parts)

    def factor(self, a, b, reuse=-1):
        lu = self.pack.factor(self.MP_id, self.LP_id, a, b, self.row_perm, self.col_perm, self.n_interior,
                              self.kl, self.ku, self.row_axes, self.col_axes, reuse=reuse,
                              real=self.real_grading)
        if not hasattr(self, "_lu_params"):
            self._lu_params = {}
        self._lu_params[lu] = (float(a), float(b))
        self._block_inverses(lu, a, b)
        return lu

    # ---- few systems: explicit inverses of the diagonal blocks ----------------------------------------------------------
    # A 2-D problem has a few hundred pencils of ~1000 rows: a sweep is a chain of n / nblocks dependent rows that so few
    # systems cannot hide (2-D Rayleigh-Benard 512 x 256: 2 x 0.39 ms of a 1.05 ms step at 0.02 of the HBM rate).  With
    # real-graded factors and the band split into independent diagonal blocks the inverse of a block is small (515^2
    # doubles = 2 MB; 1.1 GB for 256 pencils x 2 blocks) and applying it is a streaming GEMV (blockinv_solve_kernel).  The
    # inverses are formed on the device like the sphere's (core/sphere.py): the numeric band LU of csrc/ddh_ellband.hip on
    # the TRANSPOSED blocks, one unit solve per lane, so that slot s holds row s of the inverse contiguously -- the layout
    # the GEMV streams.  Same systems as the reference's per-subproblem LU (libraries/matsolvers.py:126-149).
    def _block_inverse_plan(self, any_executor=False):
        """any_executor: the host analysis alone (tests)"""
        if getattr(self, "_binv", None) is not None:
            return self._binv
        self._binv = False
        rg, n = self.real_grading, self.n_interior
        if not getattr(self, "allow_block_inverse", False):      # (initial-value solvers: many solves per factorization)
            return False
        if os.environ.get("DDH_BLOCK_INVERSE", "1") == "0" or self.nf != 1 or rg is None or getattr(self.dist, "size", 1) > 1 or n < 128:
            return False
        if not any_executor and (getattr(self.ex, "name", "") != "hip" or not hasattr(self.pack, "set_block_inverse")):
            return False
        ns = self.n_blocks if (self.n_blocks > 1 and len(set(self.block_sizes)) == 1 and os.environ.get("DDH_SPLIT_THREADS", "1") != "0") else 1
        nh = n // ns
        ncells = self.nx // 2
        nb = self.R - n
        # border rows / columns (gauge conditions) must exist for the k = 0 pencil only: that one is flagged (dense path)
        if nb and (np.any(self.row_axes[self.row_perm[n:]] & 1) or np.any(self.col_axes[self.col_perm[n:]] & 1)):
            return False
        ng = ncells * ns
        if ns * nh != n or nh > 1024 or ng * nh * nh * 8 > 2.5e9 or ng * (nh + 112) * (nh + 63) * 8 > 3e9:
            return False
        rinv, cinv = np.empty(self.R, dtype=np.int64), np.empty(self.R, dtype=np.int64)
        rinv[self.row_perm] = np.arange(self.R)
        cinv[self.col_perm] = np.arange(self.R)
        kx = np.asarray(self.pack.kx, dtype=np.float64)[:ncells]
        gmx = np.arange(ncells) + self.dist._mx_offset
        sel = []
        for mid in (rg["matM"], rg["matL"]):
            tl = self.pack.matrices[mid]
            i, j = rinv[tl.row], cinv[tl.col]
            ok = (i < n) & (j < n)
            if np.any(ok & (i // nh != j // nh)):
                return False                                  # (an entry couples two blocks: not block diagonal after all)
            sel.append((tl, i, j, ok))
        kl_b = max(int((i[ok] - j[ok]).max()) if ok.any() else 0 for (_, i, j, ok) in sel)
        ku_b = max(int((j[ok] - i[ok]).max()) if ok.any() else 0 for (_, i, j, ok) in sel)
        klT, kuT = max(ku_b, 0), max(kl_b, 0)                 # the TRANSPOSED blocks
        if klT > 35 or klT + kuT > 96:
            return False
        W = klT + kuT + 1
        bands = []
        for (tl, i, j, ok) in sel:
            B = np.zeros((ng, nh, W))
            t = np.flatnonzero(ok)
            val = tl.coef[t].real[None, :] * kx[:, None] ** tl.ex[t].astype(float)[None, :]       # [cell][term]
            val = np.where((tl.dx[t] != 0)[None, :] & (gmx != 0)[:, None], 0.0, val)
            blk, ii, jj = i[t] // nh, i[t] % nh, j[t] % nh
            g = np.arange(ncells)[:, None] * ns + blk[None, :]
            np.add.at(B, (g, np.broadcast_to(jj, g.shape), np.broadcast_to(ii - jj + klT, g.shape)), val)   # B^T[jj][ii]
            bands.append(B)
        plan = type("BlockBandPlan", (), {})()
        plan.nl, plan.nmax, plan.kl, plan.ku, plan.mp, plan.nbc = ng, nh, klT, kuT, 0, 0
        plan.n = np.full(ng, nh, dtype=np.int32)
        plan.nbc_of = np.zeros(ng, dtype=np.int32)
        plan.T = np.zeros((ng, 1, 1))
        plan.P = np.zeros((ng, nh, 1))
        plan.MB, plan.LB = bands
        plan.row_index = plan.col_index = None
        self._binv = dict(plan=plan, ns=ns, nh=nh, ng=ng, ncells=ncells, dev=None, rhs=None, x={}, n_set=None)
        return self._binv

    def _drop_block_inverses(self, bi=None):
        """Every registered explicit inverse is withdrawn from the library BEFORE its device memory is released (the
        pack holds caller-owned pointers): the sweeps of the kept LU take over."""
        bi = bi if isinstance(bi, dict) else getattr(self, "_binv", None)
        if isinstance(bi, dict):
            for k in list(bi["x"]):
                self.pack.set_block_inverse(k, None)
            bi["x"] = {}

    def _block_inverses(self, lu, a, b):
        bi = self._block_inverse_plan()
        if not bi:
            return
        ex, plan, ns, nh, ng = self.ex, bi["plan"], bi["ns"], bi["nh"], bi["ng"]
        # a re-factorization (reuse=lu) changes (a, b): whatever inverse this id holds is stale until it has been re-formed
        self.pack.set_block_inverse(lu, None)
        info = self.pack.lu_info(lu)                    # (the library's own view of the blocks must be the one planned for)
        if info["nsplit"] != ns or info["rows_per_block"] != nh or not info["real"] or info["pair"]:
            self._drop_block_inverses(bi)
            self._binv = False
            return
        flagged = sorted(getattr(self.pack, "flagged", {}).get(lu, []))
        nb = self.R - self.n_interior
        if nb:
            # Border (gauge) unknowns exist for the k = 0 pencil only (checked by the plan): blockinv_solve_kernel ignores
            # the border coupling of every pencil it solves, so that pencil must be one of the FLAGGED ones (dense path).
            # If its band block happens to be numerically regular (not flagged) the sweeps, which carry the Schur part, stay.
            k0 = np.flatnonzero(np.arange(bi["ncells"]) + self.dist._mx_offset == 0)
            if any(int(c) not in flagged for c in k0):
                return
        if bi["dev"] is None or bi["n_set"] != flagged:
            # (the flagged pencils' blocks are singular: their groups are skipped -- a plan per set of flagged cells)
            self._drop_block_inverses(bi)               # (inverses registered for the other set of cells go with their memory)
            plan.n = np.full(ng, nh, dtype=np.int32)
            for cell in flagged:
                plan.n[cell * ns:(cell + 1) * ns] = 0
            rowoff = np.broadcast_to(np.arange(nh, dtype=np.int64)[None, :] * nh, (ng, nh))
            coloff = np.arange(ng, dtype=np.int64)[:, None] * (nh * nh) + np.arange(nh, dtype=np.int64)[None, :] * nh
            try:
                bi["dev"] = ex.make_ell_band(plan, 1, nh, ng, nh, [nh] * ng, offsets=(rowoff, coloff, 1))
            except Exception as e:                       # (band wider than the compiled windows, memory): keep the sweeps
                logger.info("block inverses not available (%s): the sweeps stay" % (e,))
                self._binv = False
                return
            bi["rhs"] = ex.from_host(np.eye(nh))
            bi["n_set"] = flagged
            logger.info("LHS of %d pencils: explicit inverses of %d x %d diagonal blocks (%d rows), %.2f GB per factorization"
                        % (bi["ncells"], ng, ns, nh, ng * nh * nh * 8 / 1e9))
        if lu not in bi["x"]:
            if len(bi["x"]) >= 2:                        # (schemes with many distinct implicit coefficients: keep the sweeps)
                return
            bi["x"][lu] = (len(bi["x"]), ex.zeros((ng, nh, nh)))
        idx, x = bi["x"][lu]
        try:
            bi["dev"].factor(a, b, index=idx)
        except Exception as e:                           # (a zero pivot in a block the pencil LU handled: keep the sweeps)
            logger.warning("block inverses switched off (%s)" % (e,))
            self._drop_block_inverses(bi)
            self._binv = False
            return
        bi["dev"].solve(idx, bi["rhs"], x)
        if not self._block_inverse_residual_ok(bi, idx, x, a, b):
            self._drop_block_inverses(bi)
            self._binv = False
            return
        self.pack.set_block_inverse(lu, x)

    def _block_inverse_residual_ok(self, bi, idx, x, a, b, tol=1e-8):
        """||B^T X - I||_max on sampled blocks (first, middle, last un-flagged group; X = B^-T as the GEMV streams it), formed
        on the host from the band arrays the inverses were made of.  Every column of X is a unit solve through a band LU
        WITHOUT row interchanges inside the block, so this residual says whether that factorization was stable (measured:
        5e-12 at 128 x 64, see `solver._binv["residual"]`, printed by tools/bench_configs.py for 512 x 256); above `tol` the
        sweeps of the pivoted pencil LU stay.  The OTHER residual, ||X B^T - I||, is cond(B) eps whatever the algorithm (1e-10
        at cond 3e7) -- it bounds the forward error of x = X r exactly as cond(B) eps bounds that of a backward-stable solve
        -- and is recorded as `left_residual`, not judged.  Three blocks of <= 1024^2 doubles cross PCIe per factorization."""
        if os.environ.get("DDH_BLOCK_INVERSE_CHECK", "1") == "0":
            return True
        plan, nh = bi["plan"], bi["nh"]
        live = np.flatnonzero(plan.n > 0)
        if live.size == 0:
            return True
        worst = left = 0.0
        for g in sorted({int(live[0]), int(live[live.size // 2]), int(live[-1])}):
            band = a * plan.MB[g] + b * plan.LB[g]                      # [row of B^T][kl + (col - row)]
            Bt = np.zeros((nh, nh))
            for d in range(band.shape[1]):
                off = d - plan.kl
                i = np.arange(max(0, -off), min(nh, nh - off))
                Bt[i, i + off] = band[i, d]
            Xg = self.ex.download(x[g])                                  # the layout the GEMV streams: Xg[k][i] = B^-1[i][k]
            scale = max(1.0, float(np.abs(Bt).max()))
            worst = max(worst, float(np.abs(Bt @ Xg - np.eye(nh)).max()) / scale)
            left = max(left, float(np.abs(Xg @ Bt - np.eye(nh)).max()))
        bi["residual"], bi["left_residual"] = worst, left
        if not (worst <= tol):
            logger.warning("explicit block inverses: ||B^T X - I|| = %.1e > %.0e, the sweeps stay" % (worst, tol))
            return False
        return True


def _two_colour(n, row, col, label):
    """x_r[row] + x_c[col] = label (mod 2) for every entry; returns (x_r, x_c) or None if inconsistent."""
    from collections import deque
    adj_r = [[] for _ in range(n)]
    adj_c = [[] for _ in range(n)]
    seen = {}
    for r, c, l in zip(row.tolist(), col.tolist(), np.asarray(label).tolist()):
        key = (r, c)
        if key in seen:
            if seen[key] != l:
                return None
            continue
        seen[key] = l
        adj_r[r].append((c, l))
        adj_c[c].append((r, l))
    xr = np.full(n, -1, dtype=int)
    xc = np.full(n, -1, dtype=int)
    for start in range(n):
        if xr[start] != -1 or not adj_r[start]:
            continue
        xr[start] = 0
        dq = deque([("r", start)])
        while dq:
            kind, i = dq.popleft()
            if kind == "r":
                for c, l in adj_r[i]:
                    want = (l - xr[i]) % 2
                    if xc[c] == -1:
                        xc[c] = want
                        dq.append(("c", c))
                    elif xc[c] != want:
                        return None
            else:
                for r, l in adj_c[i]:
                    want = (l - xc[i]) % 2
                    if xr[r] == -1:
                        xr[r] = want
                        dq.append(("r", r))
                    elif xr[r] != want:
                        return None
    xr[xr == -1] = 0
    xc[xc == -1] = 0
    return xr, xc


def _matrix_times_termlist(T, tl):
    """(constant sparse matrix) @ (term list): rows mixed by T, monomials unchanged; rows T leaves alone keep their terms
    bit for bit."""
    from ..pencilpack import TermList
    if tl.nterms == 0:
        return tl
    T = T.tocsc()
    ident = np.ones(T.shape[0], dtype=bool)
    coo = T.tocoo()
    off = (coo.row != coo.col) | (coo.data != 1.0)
    ident[coo.row[off]] = False
    ident[coo.col[off]] = False
    plain = ident[tl.row]
    parts = [[tl.row[plain]], [tl.col[plain]], [tl.coef[plain]], [tl.ex[plain]], [tl.ey[plain]], [tl.dx[plain]], [tl.dy[plain]]]
    for t in np.flatnonzero(~plain):
        c = T.getcol(int(tl.row[t])).tocoo()
        n = c.nnz
        parts[0].append(c.row); parts[1].append(np.full(n, tl.col[t])); parts[2].append(c.data * tl.coef[t])
        for k, name in enumerate(("ex", "ey", "dx", "dy")):
            parts[3 + k].append(np.full(n, getattr(tl, name)[t]))
    out = TermList(tl.nrows, tl.ncols, *[np.concatenate(x) for x in parts])
    if (~plain).any():
        out = out.consolidated(0.0)
        # cancellation residue of the combinations: only in the rows T mixes, relative to each row's own largest entry (the
        # rows T leaves alone keep every term, however small against the rest of the matrix)
        rowmax = np.zeros(tl.nrows)
        np.maximum.at(rowmax, out.row, np.abs(out.coef))
        keep = ident[out.row] | (np.abs(out.coef) > 1e-14 * rowmax[out.row])
        out = TermList(out.nrows, out.ncols, out.row[keep], out.col[keep], out.coef[keep], out.ex[keep], out.ey[keep],
                       out.dx[keep], out.dy[keep])
    return out


def _termlist_times_matrix(tl, P, cutoff=1e-12):
    """(term list) @ (constant sparse matrix): same monomials, columns mixed by P.  Entries below `cutoff` RELATIVE to
    the largest entry of their monomial block are cancellation residue of the recombination and are dropped (an
    absolute threshold would also remove genuine entries of problems with tiny physical coefficients)."""
    from scipy import sparse
    from ..pencilpack import TermList
    if tl.nterms == 0:
        return tl
    key = np.stack([tl.ex, tl.ey, tl.dx, tl.dy], axis=1).astype(np.int64)
    uniq, inv = np.unique(key, axis=0, return_inverse=True)
    inv = inv.ravel()
    rows, cols, coefs, exs = [], [], [], []
    for u in range(len(uniq)):
        sel = inv == u
        A = sparse.coo_matrix((tl.coef[sel], (tl.row[sel], tl.col[sel])), shape=(tl.nrows, tl.ncols)).tocsr()
        AP = (A @ P).tocoo()
        keep = np.abs(AP.data) > cutoff * (np.abs(AP.data).max() if AP.nnz else 1.0)
        rows.append(AP.row[keep]); cols.append(AP.col[keep]); coefs.append(AP.data[keep])
        exs.append(np.tile(uniq[u], (int(keep.sum()), 1)))
    e = np.concatenate(exs)
    return TermList(tl.nrows, tl.ncols, np.concatenate(rows), np.concatenate(cols), np.concatenate(coefs),
                    e[:, 0], e[:, 1], e[:, 2], e[:, 3])


class InitialValueSolver(IVPLifecycle, SolverBase):
    """The life cycle (proceed / step / evolve / log_stats / clocks / Hermitian schedule) is the shared
    core/ivp_common.py::IVPLifecycle, i.e. the reference's core/solvers.py:594-778."""

    def __init__(self, problem, timestepper, enforce_real_cadence=100, warmup_iterations=10, **kw):
        t0 = time.time()
        SolverBase.__init__(self, problem, **kw)
        self._enable_state_tiling()
        self.sim_time_field = problem.time
        self._sim_time = 0.0
        self._init_lifecycle(enforce_real_cadence, warmup_iterations)
        if isinstance(timestepper, str):
            timestepper = ts_mod.schemes[timestepper]
        self.allow_block_inverse = True
        self.timestepper = timestepper(self)
        self.setup_time = time.time() - t0
        self.total_modes = self.R * self.nx * self.ny
        # launch-bound problems replay their fixed-timestep steps from HIP graphs by default (core/ivp_common.py) -- but only
        # when every right-hand side is built from the state alone: a parameter field (a forcing the script may rewrite
        # between steps, the time field `t`) is uploaded by ordinary launches, which a replayed graph would not see
        state_ids = {id(v) for v in self.variables}
        rhs_fields, opaque = set(), False
        for eq in self.equations:
            F = eq["F"]
            if F is None or isinstance(F, (int, float, complex)):
                continue
            if hasattr(F, "leaves"):
                rhs_fields |= {id(f) for f in F.leaves() if not getattr(f, "_is_number", False)}
            else:
                opaque = True                    # (an object whose inputs cannot be listed: not provably state-only)
        if not opaque and rhs_fields <= state_ids:
            self.step_graph_auto_modes = 1 << 22
        self.handlers = []
        from .output import OutputEvaluator
        self.evaluator = OutputEvaluator(self)       # analysis handlers: evaluated at the start of a step
        self._step_hooks = [self.evaluator.step_hook]

    @property
    def sim_time(self):
        return self._sim_time

    @sim_time.setter
    def sim_time(self, t):
        self._sim_time = float(t)
        self.sim_time_field["g"] = float(t)

    def _hermitian_round_trip(self, f):
        if self.dist.size > 1 and any(b is None for b in f.domain.by_axis):
            # tau / constant fields on several ranks: they have no distributed grid layout here, and the round trip
            # could only clear modes the solve already writes as exact zeros (store_sys, csrc/ddh_pencil.hip)
            return
        scales = f.scales
        f.require_grid_space(f.domain.dealias)
        f.require_coeff_space()
        f.change_scales(scales)              # (coefficient layout: bookkeeping only)


class LinearBoundaryValueSolver(SolverBase):
    """L.X = F in one batched factor + solve (core/solvers.py LBVP role)."""

    _lu = None

    def solve(self, rebuild_matrices=False):
        """The LHS does not change between calls: it is factored once (the reference factors at build time,
        core/solvers.py:393-406) and re-factored into the same device storage only on request."""
        self.sync_state_to_device()
        if getattr(self, "_F_lbvp", None) is None:
            self._F_lbvp = self.ex.empty((self.R, self.nx, self.ny))        # owned by the solver, reused by every solve
        F = self._F_lbvp
        self.evaluate_F(F)
        if self._lu is None or rebuild_matrices:
            self._lu = self.factor(0.0, 1.0, reuse=(-1 if self._lu is None else self._lu))
        SolverBase.solve(self, self._lu, F, self.X)
        self.mark_state_current()
