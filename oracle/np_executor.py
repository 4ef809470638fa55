"""
TEST INFRASTRUCTURE ONLY -- numpy/scipy executor with the HipExecutor interface.

It restates the reference's CPU algorithm for every device operation: scipy.fft transforms with
separate pack/scale passes (oracle/np_transforms.py), and a Python loop over pencils with scipy CSR
mat-vecs and SuperLU factorizations (oracle/np_pencil.py).  Tests inject it explicitly
(`Distributor(..., executor=NumpyExecutor())`) to check the host logic on machines without a GPU
and as the parity oracle / CPU baseline; the product never selects it by itself.
"""

import numpy as np

from . import np_pencil as npp
from . import np_transforms as npt


class _NpPack:
    def __init__(self, nf, nrows, nx, ny, kx, ky, mx_offset=0):
        self.mx_offset = mx_offset
        self.nf, self.nrows, self.nx, self.ny = nf, nrows, nx, ny
        self.kx, self.ky = np.asarray(kx, float), np.asarray(ky, float)
        self.mats = []
        self.lus = []
        self.lu_meta = {}

    def set_pairing(self, row_swap, col_swap, min_systems=0):
        """The x <-> y symmetry the solver found (dedalus_amd/core/solvers.py _build_pairing).  The oracle keeps one
        factorization per pencil -- it is the independent check of the shared ones -- and only records the maps."""
        self.pairing = None if row_swap is None else (np.asarray(row_swap), np.asarray(col_swap))

    def add_matrix(self, tl):
        self.mats.append(npp.TermList(tl.nrows, tl.ncols, tl.row, tl.col, tl.coef, tl.ex, tl.ey, tl.dx, tl.dy))
        return len(self.mats) - 1

    def matvec(self, mid, x, y):
        A = self.mats[mid]
        y[...] = npp.matvec(A, x.reshape(A.ncols, self.nx, self.ny), self.nf, self.kx, self.ky, self.mx_offset)

    def add_upper_bands(self, nz, offsets, bands):
        from scipy import sparse
        bands = np.asarray(bands, float).reshape(len(offsets), nz)
        C = sparse.diags([bands[d, :nz - o] for d, o in enumerate(offsets)], list(offsets), shape=(nz, nz)).tocsr()
        self.posts = getattr(self, "posts", [])
        self.posts.append((nz, C))
        return len(self.posts) - 1

    def matvec_solve(self, mid, bid, x, y):
        """solve_upper_sparse (tools/array.py:206-232) applied to every output component."""
        from scipy.sparse.linalg import spsolve_triangular
        self.matvec(mid, x, y)
        nz, C = self.posts[bid]
        yy = y.reshape(-1, nz, self.nx * self.ny)
        for c in range(yy.shape[0]):
            yy[c] = spsolve_triangular(C, yy[c].copy(), lower=False)

    def factor(self, matM, matL, a, b, row_perm, col_perm, n_interior, kl, ku, row_axes, col_axes, reuse=-1,
               real=None):
        lu = npp.PencilLU(self.mats[matM], self.mats[matL], a, b, self.nf, self.nx, self.ny, self.kx, self.ky,
                          np.asarray(row_axes), np.asarray(col_axes), self.mx_offset)
        if reuse >= 0:
            self.lus[reuse] = lu
            return reuse
        self.lus.append(lu)
        return len(self.lus) - 1

    def solve(self, lu_id, rhs, x):
        x[...] = self.lus[lu_id].solve(rhs).reshape(x.shape)

    MAX_RHS_TERMS = 8

    def solve_lincomb(self, lu_id, xs, alphas, x):
        """axpy chain into the RHS buffer, then the per-pencil solves (timesteppers.py:617-623, 641-643)"""
        rhs = np.zeros_like(x)
        for v, a in zip(xs, alphas):
            rhs += a * np.asarray(v).reshape(x.shape)
        self.solve(lu_id, rhs, x)

    def lu_bytes(self, lu_id):
        return 0


class NumpyExecutor:
    name = "numpy-oracle"

    def empty(self, shape):
        return np.zeros(tuple(int(s) for s in np.atleast_1d(shape)))

    zeros = empty

    def from_host(self, a):
        return np.array(a, dtype=np.float64)

    def download(self, t):
        return np.array(t)

    def upload(self, dst, a):
        dst[...] = np.asarray(a).reshape(dst.shape)

    def copy(self, dst, src):
        dst[...] = np.asarray(src).reshape(dst.shape)

    def assign(self, dst_view, src_view):
        dst_view[...] = src_view

    def fill_zero(self, a):
        a[...] = 0.0

    def sync(self):
        pass

    def lincomb(self, y, xs, alphas):
        acc = np.zeros_like(y)
        for x, a in zip(xs, alphas):
            acc += a * x.reshape(y.shape)
        y[...] = acc

    def make_scatter(self, flat_idx, vals):
        return (np.asarray(flat_idx, dtype=np.int64), np.asarray(vals, dtype=np.float64))

    def scatter_add(self, y, sparse):
        idx, vals = sparse
        y.reshape(-1)[idx] += vals

    def scatter_set(self, y, sparse):
        idx, vals = sparse
        y.reshape(-1)[idx] = vals

    def bilinear(self, out, ncomp_out, a, b, npts, terms):
        a2, b2 = a.reshape(-1, npts), b.reshape(-1, npts)
        o = np.zeros((ncomp_out, npts))
        for (ic, ia, ib, cf) in terms:
            o[ic] += cf * a2[ia] * b2[ib]
        out[...] = o.reshape(out.shape)

    FUSED_LIMITS = dict(na=3, nb=12, nc=4, terms=32, max_grid=1536)

    def fused_capable(self, spec):
        return spec[0] == "rfft" and spec[1] <= self.FUSED_LIMITS["max_grid"]

    @staticmethod
    def _fourier_diff(lines, dscale):
        """DifferentiateRealFourier (core/basis.py:1233-1260) on [nlines][M] (cos, msin) lines."""
        if not dscale:
            return lines
        M = lines.shape[1]
        k = dscale * np.arange(M // 2)
        out = np.empty_like(lines)
        out[:, 0::2] = -k * lines[:, 1::2]
        out[:, 1::2] = k * lines[:, 0::2]
        return out

    def rfft_bilinear_fused(self, spec, basis, a_list, b_list, out_list, nlines, terms, a_dscale=None,
                            b_dscale=None):
        """Unfused restatement: (derivative,) backward transforms (core/transforms.py:559-565), product
        (core/arithmetic.py:666-674), forward transform (:551-557) along the last axis."""
        N, M = spec[1], spec[2]
        a_dscale = a_dscale if a_dscale is not None else [0.0] * len(a_list)
        b_dscale = b_dscale if b_dscale is not None else [0.0] * len(b_list)
        ga = [npt.rfft_backward(self._fourier_diff(a.reshape(nlines, M), ds).reshape(nlines, M, 1), 1, N)
              for a, ds in zip(a_list, a_dscale)]
        gb = [npt.rfft_backward(self._fourier_diff(b.reshape(nlines, M), ds).reshape(nlines, M, 1), 1, N)
              for b, ds in zip(b_list, b_dscale)]
        acc = [np.zeros((nlines, N, 1)) for _ in out_list]
        for (ic, ia, ib, cf) in terms:
            acc[ic] += cf * ga[ia] * gb[ib]
        for o, g in zip(out_list, acc):
            o[...] = npt.rfft_forward(g, 1, M).reshape(o.shape)

    def transform(self, spec, basis, direction, src, dst, outer, inner, deriv=0.0):
        kind = spec[0]
        n_in = src.size // (outer * inner)
        s3 = src.reshape(outer, n_in, inner)
        if deriv:
            # DifferentiateRealFourier (core/basis.py:1233-1260) along the transformed axis
            assert kind == "rfft" and direction == "backward"
            k = (deriv * np.arange(n_in // 2))[None, :, None]
            d3 = np.empty_like(s3)
            d3[:, 0::2, :] = -k * s3[:, 1::2, :]
            d3[:, 1::2, :] = k * s3[:, 0::2, :]
            s3 = d3
        if kind == "rfft":
            N, M = spec[1], spec[2]
            res = npt.rfft_forward(s3, 1, M) if direction == "forward" else npt.rfft_backward(s3, 1, N)
        elif kind == "cheb":
            N, M = spec[1], spec[2]
            conv = None
            if basis.a != basis.a0 or basis.b != basis.b0:
                from dedalus_amd.tools import jacobi
                conv = jacobi.conversion_matrix(M, basis.a0, basis.b0, basis.a, basis.b)
            res = npt.cheb_forward(s3, 1, M, conv) if direction == "forward" else npt.cheb_backward(s3, 1, N, conv)
        elif kind == "mmt":
            fwd, bwd = basis.mmt_matrices(spec[1])
            res = npt.apply_matrix_along_axis(fwd if direction == "forward" else bwd, s3, 1)
        else:
            raise NotImplementedError(kind)
        dst[...] = res.reshape(dst.shape)

    def transform_dual(self, spec, basis, src, dst, dst_deriv, outer, inner, dscale):
        """The field's backward RealFourier transform and that of its derivative (two plain transforms here)."""
        self.transform(spec, basis, "backward", src, dst, outer, inner)
        self.transform(spec, basis, "backward", src, dst_deriv, outer, inner, deriv=dscale)

    def transform_dual_z(self, spec, basis, src, dst, dst_deriv, dvec, outer, inner):
        """Plain Chebyshev backward transform of src, and that of the superdiagonal operator dvec applied to src, taken in
        `basis` (the derivative's basis: conversion solve first, core/transforms.py:876-890)."""
        N, M = spec[1], spec[2]
        n_in = src.size // (outer * inner)
        s3 = src.reshape(outer, n_in, inner)
        dst[...] = npt.cheb_backward(s3, 1, N, None).reshape(dst.shape)
        d3 = np.zeros_like(s3)
        d3[:, :-1, :] = np.asarray(dvec)[None, :-1, None] * s3[:, 1:, :]
        from dedalus_amd.tools import jacobi
        conv = jacobi.conversion_matrix(M, basis.a0, basis.b0, basis.a, basis.b)
        dst_deriv[...] = npt.cheb_backward(d3, 1, N, conv).reshape(dst_deriv.shape)

    def cfl_max(self, u, ncomp, shape, inv_spacings, comp_axis):
        """compute_cfl_frequency (core/basis.py:6108-6111) + global max (extras/flow_tools.py:199-204)"""
        ug = np.abs(u.reshape((ncomp,) + tuple(shape)))
        f = np.zeros(tuple(shape))
        for c in range(ncomp):
            sh = [1] * len(shape)
            sh[comp_axis[c]] = -1
            f = f + ug[c] * np.asarray(inv_spacings[c]).reshape(sh)
        return float(f.max()) if f.size else 0.0

    def cfl_max_spherical(self, u, inv_h, inv_dr):
        """Spherical3DAdvectiveCFL.compute_cfl_frequency (core/basis.py:6199-6204) reduced with max"""
        return float(np.max(np.sqrt(u[0] ** 2 + u[1] ** 2) * inv_h + np.abs(u[2]) * inv_dr))

    def reduce3(self, x):
        """np.min / np.max / np.sum of the grid data (extras/flow_tools.py:32-47)"""
        x = np.asarray(x)
        return float(x.min()), float(x.max()), float(x.sum())

    def a2a_pack(self, src, dst, outer, na, nb, inner, P):
        """split_rows / split_columns of AlltoallvTranspose (core/transposes.pyx:359-445) restated"""
        s = src.reshape(outer, P, na // P, nb * inner)
        dst[...] = np.ascontiguousarray(np.transpose(s, (1, 0, 2, 3))).reshape(dst.shape)

    def a2a_unpack(self, src, dst, outer, na, nb, inner, P):
        s = src.reshape(P, outer, na, nb // P, inner)
        dst[...] = np.ascontiguousarray(np.transpose(s, (1, 2, 0, 3, 4))).reshape(dst.shape)

    def make_recombination(self, slot_map, mats):
        return (np.asarray(slot_map), np.asarray(mats), len(mats))

    def regularity_recombine(self, data, table, radial_factor_d=None):
        if radial_factor_d is not None:
            data *= np.asarray(radial_factor_d).reshape(1, 1, 1, -1)
        if table is None or data.shape[0] == 1:
            return
        sm, mats, _ = table
        for i1 in range(data.shape[1]):
            for i2 in range(data.shape[2]):
                k = int(sm[i1, i2])
                if k >= 0:
                    data[:, i1, i2, :] = mats[k] @ data[:, i1, i2, :]

    # ---- sphere coefficient-space operations (restating csrc/ddh_sphere.hip; the reference equivalents are
    # SpinRecombinationBasis.forward/backward_spin_recombination core/basis.py:1595-1663, the per-m subproblem
    # matrix products core/subsystems.py:497-596 and the per-m LHS solves libraries/matsolvers.py:126-149)
    def spin_recombine(self, src, dst, mat):
        nc, n2, inner = src.shape
        v = src.reshape(nc, n2 // 2, 2, inner).transpose(0, 2, 1, 3).reshape(2 * nc, n2 // 2, inner)
        r = np.einsum("rc,cqx->rqx", np.asarray(mat), v)
        dst[...] = r.reshape(nc, 2, n2 // 2, inner).transpose(0, 2, 1, 3).reshape(nc, n2, inner)

    def make_sphere_terms(self, nm, nl, ncomp_out, terms):
        class _Terms:
            def apply(self_, x, y):
                z = x[:, 0::2, :] + 1j * x[:, 1::2, :]
                out = np.zeros((ncomp_out, nm, nl), dtype=complex)
                for (co, ci, d, coef) in terms:
                    sh = np.zeros((nm, nl), dtype=complex)
                    if d >= 0:
                        sh[:, :nl - d] = z[ci][:, d:]
                    else:
                        sh[:, -d:] = z[ci][:, :nl + d]
                    out[co] += coef * sh
                y[:, 0::2, :] = out.real
                y[:, 1::2, :] = out.imag
        return _Terms()

    def make_ell_terms(self, nm, nl, nr, ncomp_out, terms, slot_map=None):
        """SphericalEllOperator.operate (core/operators.py:3132-3160): per slot its radial matrix applied along n"""
        if slot_map is None:
            i1, ell = np.indices((2 * nm, nl))
            slot_map = np.where(i1 // 2 <= ell, ell, -1)
        sm = np.asarray(slot_map)
        live = sm >= 0

        class _Terms:
            def apply(self_, x, y):
                out = np.zeros((ncomp_out, 2 * nm, nl, nr))
                for (co, ci, mats) in terms:
                    per_slot = mats[np.where(live, sm, 0)]                  # [2 nm][nl][nr][nr]
                    out[co] += np.einsum("alij,alj->ali", per_slot, x[ci])
                y[...] = out * live[None, :, :, None]
        return _Terms()

    def make_cgemv_batch(self, nm, nl, ncomp, mats):
        mats = [np.asarray(a, dtype=complex) for a in mats]

        class _Batch:
            nbytes = sum(a.nbytes for a in mats)

            def apply(self_, x, y):
                y[...] = 0.0
                for m in range(nm):
                    ne = nl - m
                    if ne <= 0:
                        continue
                    z = (x[:, 2 * m, m:] + 1j * x[:, 2 * m + 1, m:]).reshape(-1)
                    r = (mats[m] @ z).reshape(ncomp, ne)
                    y[:, 2 * m, m:] = r.real
                    y[:, 2 * m + 1, m:] = r.imag
        return _Batch()

    def make_grouped_mmt(self, n_grid, groups, ms, fwd_mats, bwd_mats):
        from . import np_swsh

        class _Plan:
            def forward(self_, g, c):
                np_swsh.forward_reduced(g, c, groups, {int(m): a for m, a in zip(ms, fwd_mats)})

            def backward(self_, c, g):
                np_swsh.backward_reduced(c, g, groups, {int(m): a for m, a in zip(ms, bwd_mats)})
        return _Plan()

    def make_dense_inverse(self, Ms, Ls, row_valid, col_valid, complex_=False):
        """The reference's per-subproblem solver restated as explicit inverses of the valid blocks
        (libraries/matsolvers.py:126-149 on the matrices of core/subsystems.py:497-596, valid modes :540-556)."""
        Ms = [np.asarray(a) for a in Ms]
        Ls = [np.asarray(a) for a in Ls]

        class _Inv:
            def compute(self_, a, b):
                outs = []
                for M, L, rv, cv in zip(Ms, Ls, row_valid, col_valid):
                    rv, cv = np.asarray(rv, dtype=bool).ravel(), np.asarray(cv, dtype=bool).ravel()
                    inv = np.zeros(M.shape, dtype=(complex if complex_ else float))
                    if rv.any():
                        A = a * M + b * L
                        if complex_:
                            import scipy.sparse as sp
                            import scipy.sparse.linalg as spla
                            sub = sp.csc_matrix(A[np.ix_(rv, cv)])
                            inv[np.ix_(cv, rv)] = spla.splu(sub).solve(np.eye(sub.shape[0], dtype=complex))
                        else:
                            inv[np.ix_(cv, rv)] = np.linalg.inv(A[np.ix_(rv, cv)])
                    outs.append(inv)
                return outs
        return _Inv()

    def make_cgemv_batch_flat(self, nm, nl, ncomp, mats, old=None):
        return self.make_cgemv_batch(nm, nl, ncomp, mats)

    def make_ell_terms_from_dense(self, nm, nl, nr, ncomp, mats, old=None):
        inv = np.array(mats)
        blocks = []
        for co in range(ncomp):
            for ci in range(ncomp):
                blk = inv[:, co * nr:(co + 1) * nr, ci * nr:(ci + 1) * nr]
                if np.any(blk != 0):
                    blocks.append((co, ci, np.ascontiguousarray(blk)))
        return self.make_ell_terms(nm, nl, nr, ncomp, blocks)

    def make_pack(self, nf, nrows, nx, ny, kx, ky, mx_offset=0):
        return _NpPack(nf, nrows, nx, ny, kx, ky, mx_offset)
