"""
Host wrapper of the batched pencil engine in libdedalus_hip.so (ddh_pencil_*).

A PencilPack owns, for ALL pencils of a problem at once: the cell geometry, the matrices as
polynomial term lists, and the bordered-band LU factorizations.  It replaces the reference's
per-subproblem Python loops (core/timesteppers.py:588-591, 611-614, 630-643) and per-pencil
scipy/SuperLU objects (core/subsystems.py:497-596, libraries/matsolvers.py:126-149).
"""

import ctypes as C

import numpy as np

from . import libhip
from .device import ptr


class TermList:
    """Sparse matrix whose entries are polynomials in the pencil wavenumbers:
        A[row, col] += coef * kx^ex * ky^ey * [mx==0]^dx * [my==0]^dy
    """

    __slots__ = ("nrows", "ncols", "row", "col", "coef", "ex", "ey", "dx", "dy")

    def __init__(self, nrows, ncols, row=(), col=(), coef=(), ex=(), ey=(), dx=(), dy=()):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.row = np.asarray(row, dtype=np.int32).ravel()
        self.col = np.asarray(col, dtype=np.int32).ravel()
        self.coef = np.asarray(coef, dtype=np.complex128).ravel()
        n = self.row.size
        self.ex = np.asarray(ex, dtype=np.int8).ravel() if np.size(ex) else np.zeros(n, np.int8)
        self.ey = np.asarray(ey, dtype=np.int8).ravel() if np.size(ey) else np.zeros(n, np.int8)
        self.dx = np.asarray(dx, dtype=np.int8).ravel() if np.size(dx) else np.zeros(n, np.int8)
        self.dy = np.asarray(dy, dtype=np.int8).ravel() if np.size(dy) else np.zeros(n, np.int8)

    @property
    def nterms(self):
        return self.row.size

    def consolidated(self, cutoff=0.0):
        """Merge duplicate (row, col, monomial) terms and drop |coef| <= cutoff."""
        if self.nterms == 0:
            return self
        key = np.stack([self.row.astype(np.int64), self.col.astype(np.int64), self.ex.astype(np.int64),
                        self.ey.astype(np.int64), self.dx.astype(np.int64), self.dy.astype(np.int64)], axis=1)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        coef = np.zeros(len(uniq), dtype=np.complex128)
        np.add.at(coef, inv.ravel(), self.coef)
        keep = np.abs(coef) > cutoff
        u = uniq[keep]
        return TermList(self.nrows, self.ncols, u[:, 0], u[:, 1], coef[keep], u[:, 2], u[:, 3], u[:, 4], u[:, 5])

    def dense(self, kx, ky, mx, my, sign=1):
        """Dense complex matrix of one system (host-side; used for the few flagged pencils)."""
        val = self.coef * (sign * kx) ** self.ex.astype(float) * ky ** self.ey.astype(float)
        if mx != 0:
            val = np.where(self.dx != 0, 0.0, val)
        if my != 0:
            val = np.where(self.dy != 0, 0.0, val)
        A = np.zeros((self.nrows, self.ncols), dtype=np.complex128)
        np.add.at(A, (self.row, self.col), val)
        return A


class PencilPack:
    def __init__(self, dev, nfourier, nrows, nx, ny, kx, ky, mx_offset=0):
        self.dev = dev
        self.mx_offset = int(mx_offset)
        self.nf, self.nrows, self.nx, self.ny = int(nfourier), int(nrows), int(nx), int(ny)
        self.kx = np.ascontiguousarray(kx, dtype=np.float64)
        self.ky = np.ascontiguousarray(ky, dtype=np.float64)
        geom = libhip.PencilGeom(self.nf, self.nrows, self.nx, self.ny, libhip.as_dp(self.kx), libhip.as_dp(self.ky),
                                 self.mx_offset)
        self.handle = C.c_uint64(0)
        libhip.call("ddh_pencil_create", C.byref(self.handle), C.byref(geom))
        self.S = 2 if self.nf == 2 else 1
        self.ncx = self.nx // 2 if self.nf >= 1 else 1
        self.ncy = self.ny // 2 if self.nf == 2 else 1
        self.matrices = []
        self.lu_meta = {}
        self.executor = None

    def add_matrix(self, tl):
        row = np.ascontiguousarray(tl.row)
        col = np.ascontiguousarray(tl.col)
        cre = np.ascontiguousarray(tl.coef.real)
        cim = np.ascontiguousarray(tl.coef.imag)
        ex, ey, dx, dy = (np.ascontiguousarray(a) for a in (tl.ex, tl.ey, tl.dx, tl.dy))
        pm = libhip.PolyMat(tl.nterms, libhip.as_ip(row), libhip.as_ip(col), libhip.as_dp(cre), libhip.as_dp(cim),
                            libhip.as_bp(ex), libhip.as_bp(ey), libhip.as_bp(dx), libhip.as_bp(dy))
        mid = C.c_int(-1)
        libhip.call("ddh_pencil_add_matrix", self.handle, C.byref(pm), tl.nrows, C.byref(mid))
        self.matrices.append(tl)
        return mid.value

    def _timer(self):
        return getattr(self.executor, "timer", None) if self.executor is not None else None

    def matvec(self, mat_id, x, y, owned=False, tiled=False):
        """y = A x.  owned: y is a zero-initialised buffer that only this product writes (a timestepper's M.X vector): rows
        without terms may then stay untouched (ddh_pencil_matvec_update).  tiled (with owned): y is written in the
        tile-major layout the sweeps read contiguously (ddh_pencil_matvec_update_tiled)."""
        t = self._timer()
        if t is not None:
            return t.run("pencil_matvec", (x.numel() + y.numel()) * 8, self._matvec, mat_id, x, y, owned, tiled)
        return self._matvec(mat_id, x, y, owned, tiled)

    def _matvec(self, mat_id, x, y, owned=False, tiled=False):
        if tiled:
            libhip.call("ddh_pencil_matvec_update_tiled", self.handle, mat_id, ptr(x), ptr(y), self.dev.stream)
            return
        libhip.call("ddh_pencil_matvec_update" if owned else "ddh_pencil_matvec", self.handle, mat_id, ptr(x), ptr(y),
                    self.dev.stream)

    supports_tiled_rhs = True

    def set_state_tiled(self, on):
        """The vector every solve of this pack writes and every mat-vec of it reads (the solver's state X) is tile-major
        from now on (ddh_pencil_set_state_tiled)."""
        libhip.call("ddh_pencil_set_state_tiled", self.handle, int(on))      # 0 natural, 1 tile-major rows, 2 kx-band-major
        self.state_tiled = int(on)

    def add_upper_bands(self, nz, offsets, bands):
        offs = np.ascontiguousarray(offsets, dtype=np.int32)
        b = np.ascontiguousarray(bands, dtype=np.float64)
        bid = C.c_int(-1)
        libhip.call("ddh_pencil_add_upper_bands", self.handle, int(nz), len(offs), libhip.as_ip(offs), libhip.as_dp(b),
                    C.byref(bid))
        return bid.value

    def matvec_solve(self, mat_id, bands_id, x, y):
        t = self._timer()
        if t is not None:
            return t.run("pencil_matvec", (x.numel() + y.numel()) * 8, self._matvec_solve, mat_id, bands_id, x, y)
        return self._matvec_solve(mat_id, bands_id, x, y)

    def _matvec_solve(self, mat_id, bands_id, x, y):
        libhip.call("ddh_pencil_matvec_solve", self.handle, mat_id, bands_id, ptr(x), ptr(y), self.dev.stream)

    def factor(self, matM, matL, a, b, row_perm, col_perm, n_interior, kl, ku, row_axes, col_axes, reuse=-1,
               real=None):
        """Factor a*M + b*L for every pencil; returns the LU id.
        real = dict(matM, matL, row_code, col_code) selects the real graded factorization
        (ddh_pencil_factor_real); matM / matL (complex term lists) are then only used for the dense
        inverses of flagged pencils."""
        row_perm = np.ascontiguousarray(row_perm, dtype=np.int32)
        col_perm = np.ascontiguousarray(col_perm, dtype=np.int32)
        # masks are passed in LOGICAL order
        ra = np.ascontiguousarray(np.asarray(row_axes, dtype=np.uint8)[row_perm])
        ca = np.ascontiguousarray(np.asarray(col_axes, dtype=np.uint8)[col_perm])
        lu = C.c_int(-1)
        if real is not None:
            rc = np.ascontiguousarray(np.asarray(real["row_code"], dtype=np.uint8)[row_perm])
            cc = np.ascontiguousarray(np.asarray(real["col_code"], dtype=np.uint8)[col_perm])
            libhip.call("ddh_pencil_factor_real", self.handle, real["matM"], real["matL"], float(a), float(b),
                        libhip.as_ip(row_perm), libhip.as_ip(col_perm), int(n_interior), int(kl), int(ku),
                        libhip.as_ubp(ra), libhip.as_ubp(ca), libhip.as_ubp(rc), libhip.as_ubp(cc),
                        int(reuse), C.byref(lu), self.dev.stream)
        else:
            libhip.call("ddh_pencil_factor", self.handle, matM, matL, float(a), float(b), libhip.as_ip(row_perm),
                        libhip.as_ip(col_perm), int(n_interior), int(kl), int(ku), libhip.as_ubp(ra),
                        libhip.as_ubp(ca), int(reuse), C.byref(lu), self.dev.stream)
        lu_id = lu.value
        # pencils whose band block is singular: explicit dense inverse built here on the host
        count = C.c_int(0)
        cells = np.zeros(4096, dtype=np.int64)
        libhip.call("ddh_pencil_flagged", self.handle, lu_id, C.byref(count),
                    cells.ctypes.data_as(C.POINTER(C.c_long)), cells.size)
        nflag = count.value
        self.flagged = {} if not hasattr(self, "flagged") else self.flagged
        self.flagged[lu_id] = [int(c) for c in cells[:min(nflag, cells.size)]]
        if nflag > cells.size or nflag * self.S * self.nrows ** 2 * 16 > 8e9:
            raise libhip.DdhError("%d pencils have a singular band block: problem structure unsupported by the "
                                  "bordered-band solver" % nflag)
        if nflag:
            self._flagged_inverses(lu_id, matM, matL, a, b, row_perm, col_perm, ra, ca, cells[:nflag], n_interior)
        self.lu_meta[lu_id] = dict(nflag=nflag, a=a, b=b)
        return lu_id

    def _bordered_band(self, Md, Ld, n_interior):
        """-> BorderedBandInverse for a real pencil matrix (permuted order) whose band block is singular only through ONE
        vanishing column and that has a 1 x 1 border (the k = 0 pencil of a problem with a pressure gauge: the constant
        pressure mode drops out of every interior equation and `integ(p) = 0` / tau_p close the system), else None."""
        N, n = Md.shape[0], int(n_interior)
        if N - n != 1 or n < 2:
            return None
        absB = np.abs(Md[:n, :n]) + np.abs(Ld[:n, :n])
        zc = np.flatnonzero(absB.sum(axis=0) == 0.0)
        if len(zc) != 1:
            return None
        j0 = int(zc[0])
        if Md[n, j0] == 0.0 and Ld[n, j0] == 0.0:
            return None                                   # the gauge row does not see the free mode
        from .executor import BorderedBandInverse, HipExecutor
        ex = self.executor if self.executor is not None else HipExecutor(self.dev)
        try:
            return BorderedBandInverse(ex, Md, Ld, n, j0)
        except libhip.DdhError:
            return None                                   # (band wider than the compiled windows)

    def _flagged_inverses(self, lu_id, matM, matL, a, b, row_perm, col_perm, ra, ca, cells, n_interior=None):
        """Explicit inverses of the flagged pencils (for Rayleigh-Benard: the mean mode, 1289 x 1289), formed and inverted
        ON THE DEVICE (ddh_dense_inverse_*: M and L of those pencils are uploaded once, a change of the timestep costs no
        host linear algebra -- numpy.linalg.inv of that one matrix was 204 of the 334 ms of a refactorization) and handed
        to the pack with ddh_pencil_set_dense_inverse_dev.  Rows / columns that do not exist for a pencil are masked out
        (their unknowns are exactly zero), as the reference's valid-mode filtering does (core/subsystems.py:540-556)."""
        import os
        N, S = self.nrows, self.S
        if n_interior is None:
            n_interior = N
        key = (matM, matL, row_perm.tobytes(), col_perm.tobytes(), tuple(int(c) for c in cells))
        cache = self.__dict__.setdefault("_flag_dinv", {})
        if key not in cache:
            M, L = self.matrices[matM], self.matrices[matL]
            Ms, Ls, rvs, cvs, slots = [], [], [], [], []
            for f, cell in enumerate(int(c) for c in cells):
                mx, my = (cell // self.ncy, cell % self.ncy) if self.nf == 2 else (cell, 0)
                kxv = self.kx[mx] if self.nf >= 1 else 0.0
                kyv = self.ky[my] if self.nf == 2 else 0.0
                gmx = mx + self.mx_offset
                rv = np.array([_valid(ra[i], gmx, my, self.nf) for i in range(N)], dtype=bool)
                cv = np.array([_valid(ca[i], gmx, my, self.nf) for i in range(N)], dtype=bool)
                for sidx in range(S):
                    if sidx == 1 and kxv == 0.0:
                        slots[-1].append(f * S + 1)                  # lambda(-kx) == lambda(kx) at kx = 0
                        continue
                    sign = 1 if sidx == 0 else -1
                    Ms.append(M.dense(kxv, kyv, gmx, my, sign)[np.ix_(row_perm, col_perm)])
                    Ls.append(L.dense(kxv, kyv, gmx, my, sign)[np.ix_(row_perm, col_perm)])
                    rvs.append(rv)
                    cvs.append(cv)
                    slots.append([f * S + sidx])
            # (the mean-mode pencil of a real operator is a real matrix: a real inversion is 3-4x cheaper)
            cx = any(m.imag.any() for m in Ms) or any(m.imag.any() for m in Ls)
            if not cx:
                Ms, Ls = [m.real.copy() for m in Ms], [m.real.copy() for m in Ls]
            band = None
            if (not cx and os.environ.get("DDH_FLAG_DENSE", "0") != "1" and not getattr(self, "_flag_dense", False)
                    and all(rv.all() and cv.all() for rv, cv in zip(rvs, cvs))):
                band = [self._bordered_band(Md, Ld, n_interior) for Md, Ld in zip(Ms, Ls)]
                if any(bd is None for bd in band):
                    band = None
            if len(cache) > 4:                       # (either kind of entry holds O(n^2) doubles on the device)
                cache.clear()
            if band is not None:
                cache[key] = ("band", band, slots)
            else:
                from .executor import DenseInverse, HipExecutor
                ex = self.executor if self.executor is not None else HipExecutor(self.dev)
                cache[key] = ("dev", DenseInverse(ex, Ms, Ls, rvs, cvs, cx), [m.shape[0] for m in Ms], slots, cx)
        ent = cache[key]
        if ent[0] == "band":
            for bd, sl in zip(ent[1], ent[2]):
                inv = bd.compute(a, b)
                if inv is None:                      # (a zero pivot: never seen; the dense path takes over for good)
                    self._flag_dense = True          # (this pack only: other solvers of the process keep the band path)
                    cache.pop(key)
                    return self._flagged_inverses(lu_id, matM, matL, a, b, row_perm, col_perm, ra, ca, cells, n_interior)
                for t in sl:
                    libhip.call("ddh_pencil_set_dense_inverse_dev", self.handle, lu_id, int(t), C.c_void_p(inv.data_ptr()), 0,
                                self.dev.stream)
            return
        _, dinv, sizes, slots, cx = ent
        flat = dinv.compute(a, b)
        off = 0
        for n, sl in zip(sizes, slots):
            for t in sl:
                libhip.call("ddh_pencil_set_dense_inverse_dev", self.handle, lu_id, int(t),
                            C.c_void_p(flat.data_ptr() + off * 8), int(cx), self.dev.stream)
            off += n * n * (2 if cx else 1)

    def set_block_inverse(self, lu_id, binv):
        """explicit transposed inverses [cell][block][k][i] of the diagonal blocks (device array, kept alive by the caller),
        or None: ddh_pencil_solve_recombined* then applies them instead of running the sweeps"""
        libhip.call("ddh_pencil_set_block_inverse", self.handle, int(lu_id), ptr(binv) if binv is not None else None)
        self.__dict__.setdefault("_binv_bytes", {})[int(lu_id)] = int(binv.numel()) * 8 if binv is not None else 0

    def solve(self, lu_id, rhs, x):
        t = self._timer()
        if t is not None:
            nb = self.lu_bytes(lu_id) + (rhs.numel() + x.numel()) * 8
            return t.run("pencil_solve", nb, self._solve, lu_id, rhs, x)
        return self._solve(lu_id, rhs, x)

    def _solve(self, lu_id, rhs, x):
        libhip.call("ddh_pencil_solve", self.handle, lu_id, ptr(rhs), ptr(x), self.dev.stream)

    def solve_lincomb(self, lu_id, xs, alphas, x):
        """x = (a M + b L)^-1 (sum_t alphas[t] xs[t]): the right-hand-side combination is formed inside the forward sweep
        (ddh_pencil_solve_lincomb)."""
        t = self._timer()
        if t is not None:
            nb = self.lu_bytes(lu_id) + (sum(v.numel() for v in xs) + x.numel()) * 8
            return t.run("pencil_solve", nb, self._solve_lincomb, lu_id, xs, alphas, x)
        return self._solve_lincomb(lu_id, xs, alphas, x)

    def _solve_lincomb(self, lu_id, xs, alphas, x):
        arr = (C.c_void_p * len(xs))(*[C.c_void_p(v.data_ptr()) for v in xs])
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        libhip.call("ddh_pencil_solve_lincomb", self.handle, lu_id, len(xs), arr, libhip.as_dp(al), ptr(x), self.dev.stream)

    supports_zero_rows = True

    def solve_recombined(self, lu_id, xs, alphas, p_mat_id, work, x, zero_rows=None, skip_rows=None, tiled=False):
        """x = P (a M + b L P)^-1 (sum_t alphas[t] xs[t]) (ddh_pencil_solve_recombined): recombination fused into the
        backward sweep where the kernel variant allows, else through `work` and a mat-vec.
        zero_rows = (device uint8 mask, fraction set): rows that are zero in every term; skip_rows = (mask, fraction):
        unknowns the caller does not need (ddh_pencil_solve_recombined_sparse)."""
        t = self._timer()
        if t is not None:
            info = self.lu_info(lu_id)          # masked rows are not read / written -- by the kernels that honour the masks
            read = 1.0 - (zero_rows[1] if (zero_rows is not None and info["forward"] == "lean") else 0.0)
            wrote = 1.0 - (skip_rows[1] if (skip_rows is not None and info["backward_lanes"] == 0 and info["real"]) else 0.0)
            nb = self.lu_bytes(lu_id) + (sum(v.numel() for v in xs) * read + x.numel() * wrote) * 8
            binv = getattr(self, "_binv_bytes", {}).get(lu_id)
            if binv:                            # explicit block inverses: they are what the solve streams, not the factors
                nb = binv + (sum(v.numel() for v in xs) + 3 * x.numel()) * 8      # (+ work written, read by the P mat-vec, x written)
            return t.run("pencil_solve", nb, self._solve_recombined, lu_id, xs, alphas, p_mat_id, work, x, zero_rows,
                         skip_rows, tiled)
        return self._solve_recombined(lu_id, xs, alphas, p_mat_id, work, x, zero_rows, skip_rows, tiled)

    def _solve_recombined(self, lu_id, xs, alphas, p_mat_id, work, x, zero_rows=None, skip_rows=None, tiled=False):
        arr = (C.c_void_p * len(xs))(*[C.c_void_p(v.data_ptr()) for v in xs])
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        if zero_rows is not None or skip_rows is not None or tiled:
            zp = C.c_void_p(zero_rows[0].data_ptr()) if zero_rows is not None else None
            sp = C.c_void_p(skip_rows[0].data_ptr()) if skip_rows is not None else None
            libhip.call("ddh_pencil_solve_recombined_tiled" if tiled else "ddh_pencil_solve_recombined_sparse", self.handle,
                        lu_id, len(xs), arr, libhip.as_dp(al), int(p_mat_id), ptr(work), ptr(x), zp, sp, self.dev.stream)
            return
        libhip.call("ddh_pencil_solve_recombined", self.handle, lu_id, len(xs), arr, libhip.as_dp(al), int(p_mat_id),
                    ptr(work), ptr(x), self.dev.stream)

    MAX_RHS_TERMS = 8

    def set_solve_variant(self, mode=1, fwd=-1, backward_lanes=-1):
        """Sweep variant of solve(): mode 1 by the number of systems (default), 0 one thread per system, 2 cooperative;
        fwd (0 / 1) and backward_lanes (0 / 4 / 16) override the two sweeps individually."""
        libhip.call("ddh_pencil_set_solve_variant", self.handle, int(mode), int(fwd), int(backward_lanes))
        self.variant_epoch = getattr(self, "variant_epoch", 0) + 1     # (layout decisions that depend on the variant: timesteppers)

    def set_pairing(self, row_swap, col_swap, min_systems=0):
        """Partner pencils (ddh_pencil_set_pairing): (my, mx) is solved with the factorization of (mx, my) through the
        physical row / column involutions of the x <-> y symmetry.  None, None switches it off."""
        if row_swap is None or col_swap is None:
            libhip.call("ddh_pencil_set_pairing", self.handle, None, None, 0)
            return
        rs = np.ascontiguousarray(row_swap, dtype=np.int32)
        cs = np.ascontiguousarray(col_swap, dtype=np.int32)
        libhip.call("ddh_pencil_set_pairing", self.handle, libhip.as_ip(rs), libhip.as_ip(cs), int(min_systems))

    def set_row_blocks(self, nblocks):
        """Independent diagonal blocks of equal size in the band block of later factorizations (ddh_pencil_set_row_blocks)."""
        libhip.call("ddh_pencil_set_row_blocks", self.handle, int(nblocks))

    def lu_info(self, lu_id):
        """Shape of a factorization and the sweep variant its solves launch (ddh_pencil_lu_info); cached per id and variant."""
        v = np.zeros(12, dtype=np.int32)
        libhip.call("ddh_pencil_lu_info", self.handle, lu_id, libhip.as_ip(v))
        keys = ("n", "nb", "kl", "ku", "W", "BW", "nsplit", "rows_per_block", "forward", "backward_lanes", "pair", "real")
        d = dict(zip(keys, (int(x) for x in v)))
        d["forward"] = ("general", "lean", "cooperative")[d["forward"]]
        return d

    def lu_bytes(self, lu_id):
        n = C.c_size_t(0)
        libhip.call("ddh_pencil_lu_bytes", self.handle, lu_id, C.byref(n))
        return n.value


def _valid(bits, mx, my, nf):
    if nf >= 1 and mx != 0 and not (bits & 1):
        return False
    if nf == 2 and my != 0 and not (bits & 2):
        return False
    return True
