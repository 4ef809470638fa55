"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's per-pencil linear algebra:
a Python loop over pencils, one scipy sparse matrix per pencil, SuperLU factor + solve
(core/timesteppers.py:588-591, 630-643; libraries/matsolvers.py:126-149; tools/array.py:171-203).

Works on the same "term list" matrix description and the same real [rows][nx][ny] system-vector
layout the HIP path uses, so results can be compared entry by entry.
"""

import numpy as np
from scipy import sparse
from scipy.sparse import linalg as spla


class TermList:
    """A[row, col] += coef * kx^ex * ky^ey * [mx==0]^dx * [my==0]^dy"""

    def __init__(self, nrows, ncols, row, col, coef, ex, ey, dx, dy):
        self.nrows, self.ncols = nrows, ncols
        self.row = np.asarray(row, dtype=np.int32)
        self.col = np.asarray(col, dtype=np.int32)
        self.coef = np.asarray(coef, dtype=np.complex128)
        self.ex = np.asarray(ex, dtype=np.int8)
        self.ey = np.asarray(ey, dtype=np.int8)
        self.dx = np.asarray(dx, dtype=np.int8)
        self.dy = np.asarray(dy, dtype=np.int8)

    def cached_matrix(self, kx, ky, mx, my, sign=1):
        """The reference builds every subproblem's matrices once and keeps them (core/subsystems.py:497-596)."""
        cache = self.__dict__.setdefault("_cache", {})
        key = (float(kx), float(ky), int(mx), int(my), int(sign))
        if key not in cache:
            cache[key] = self.matrix(kx, ky, mx, my, sign)
        return cache[key]

    def matrix(self, kx, ky, mx, my, sign=1):
        """scipy CSR complex matrix of one system: sign=+1 -> (kx, ky), sign=-1 -> (-kx, ky)."""
        val = self.coef * (sign * kx) ** self.ex.astype(float) * ky ** self.ey.astype(float)
        if mx != 0:
            val = np.where(self.dx != 0, 0.0, val)
        if my != 0:
            val = np.where(self.dy != 0, 0.0, val)
        return sparse.coo_matrix((val, (self.row, self.col)), shape=(self.nrows, self.ncols)).tocsr()


def cell_to_systems(v, nf, mx, my):
    """v: real [rows][nx][ny] -> list of complex vectors (one per system of the cell)."""
    if nf == 2:
        cc, cs = v[:, 2 * mx, 2 * my], v[:, 2 * mx, 2 * my + 1]
        sc, ss = v[:, 2 * mx + 1, 2 * my], v[:, 2 * mx + 1, 2 * my + 1]
        return [(cc - ss) + 1j * (cs + sc), (cc + ss) + 1j * (cs - sc)]
    if nf == 1:
        return [v[:, 2 * mx, 0] + 1j * v[:, 2 * mx + 1, 0]]
    return [v[:, 0, 0].astype(complex)]


def systems_to_cell(out, sysvals, nf, mx, my):
    if nf == 2:
        P, Q = sysvals
        out[:, 2 * mx, 2 * my] = 0.5 * (P.real + Q.real)
        out[:, 2 * mx, 2 * my + 1] = 0.5 * (P.imag + Q.imag)
        out[:, 2 * mx + 1, 2 * my] = 0.5 * (P.imag - Q.imag)
        out[:, 2 * mx + 1, 2 * my + 1] = 0.5 * (Q.real - P.real)
    elif nf == 1:
        out[:, 2 * mx, 0] = sysvals[0].real
        out[:, 2 * mx + 1, 0] = sysvals[0].imag
    else:
        out[:, 0, 0] = sysvals[0].real


def _cells(nf, nx, ny):
    ncx = nx // 2 if nf >= 1 else 1
    ncy = ny // 2 if nf == 2 else 1
    for mx in range(ncx):
        for my in range(ncy):
            yield mx, my


def matvec(A, x, nf, kx, ky, mx_offset=0):
    """y = A x for every pencil, x real [ncols][nx][ny] -> y real [nrows][nx][ny]"""
    x = x.reshape(x.shape[0], x.shape[1], -1)
    nx, ny = x.shape[1], x.shape[2]
    y = np.zeros((A.nrows, nx, ny))
    for mx, my in _cells(nf, nx, ny):
        xs = cell_to_systems(x, nf, mx, my)
        kxv = kx[mx] if nf >= 1 else 0.0
        kyv = ky[my] if nf == 2 else 0.0
        ys = [A.cached_matrix(kxv, kyv, mx + mx_offset, my, sign=(1 if s == 0 else -1)) @ xs[s] for s in range(len(xs))]
        systems_to_cell(y, ys, nf, mx, my)
    return y


def border_identity(nrows, row_axes, col_axes, mx, my, nf):
    """Identity entries pairing the k-th invalid row with the k-th invalid column of a pencil
    (the reference simply drops them: valid-mode filtering, core/subsystems.py:540-556)."""
    def valid(bits):
        if nf >= 1 and mx != 0 and not (bits & 1):
            return False
        if nf == 2 and my != 0 and not (bits & 2):
            return False
        return True
    bad_r = [r for r in range(nrows) if not valid(row_axes[r])]
    bad_c = [c for c in range(nrows) if not valid(col_axes[c])]
    assert len(bad_r) == len(bad_c), "invalid rows/cols must pair up"
    return sparse.coo_matrix((np.ones(len(bad_r)), (bad_r, bad_c)), shape=(nrows, nrows)).tocsr()


class PencilLU:
    """One SuperLU factorization per system, like the reference's LHS_solvers per subproblem."""

    def __init__(self, M, L, a, b, nf, nx, ny, kx, ky, row_axes=None, col_axes=None, mx_offset=0):
        self.nf, self.nx, self.ny = nf, nx, ny
        self.lus = {}
        N = M.nrows
        if row_axes is None:
            row_axes = np.full(N, 3, dtype=np.uint8)
            col_axes = np.full(N, 3, dtype=np.uint8)
        for mx, my in _cells(nf, nx, ny):
            kxv = kx[mx] if nf >= 1 else 0.0
            kyv = ky[my] if nf == 2 else 0.0
            g = mx + mx_offset
            ident = border_identity(N, row_axes, col_axes, g, my, nf)
            for s in range(2 if nf == 2 else 1):
                sign = 1 if s == 0 else -1
                A = (a * M.matrix(kxv, kyv, g, my, sign) + b * L.matrix(kxv, kyv, g, my, sign) + ident).tocsc()
                self.lus[(mx, my, s)] = spla.splu(A)

    def solve(self, rhs):
        rhs = rhs.reshape(rhs.shape[0], self.nx, -1)
        x = np.zeros_like(rhs)
        for mx, my in _cells(self.nf, self.nx, self.ny):
            rs = cell_to_systems(rhs, self.nf, mx, my)
            xs = [self.lus[(mx, my, s)].solve(rs[s]) for s in range(len(rs))]
            systems_to_cell(x, xs, self.nf, mx, my)
        return x
