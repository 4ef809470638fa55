"""
Linear operators on coefficient data as sums of separable terms.

Every linear Dedalus operator on a Cartesian Fourier^n x Jacobi domain is, per Fourier mode, a
polynomial in the wavenumbers times a sparse matrix along the Jacobi axis times a map between tensor
components.  A `LinExpr` stores an expression that is linear in a set of leaf fields as

    out[co, :] += coef * (i kx)^.. (i ky)^..  [mx==0]^dx [my==0]^dy  *  Z @ leaf[ci, :]

i.e. exactly what the reference builds numerically pencil by pencil in
Operator.expression_matrices / subproblem_matrix (core/operators.py:764-780, 925-946;
core/subsystems.py:497-596) -- but symbolically in the wavenumbers, once for all pencils.
"""

import numpy as np
from scipy import sparse

from ..pencilpack import TermList


class Term:
    __slots__ = ("co", "ci", "Z", "coef", "ex", "ey", "dx", "dy", "dt")

    def __init__(self, co, ci, Z, coef=1.0, ex=0, ey=0, dx=0, dy=0, dt=0):
        self.co, self.ci, self.Z, self.coef = co, ci, Z, complex(coef)
        self.ex, self.ey, self.dx, self.dy, self.dt = ex, ey, dx, dy, dt

    def copy(self, **kw):
        t = Term(self.co, self.ci, self.Z, self.coef, self.ex, self.ey, self.dx, self.dy, self.dt)
        for k, v in kw.items():
            setattr(t, k, v)
        return t


class LinExpr:
    """Expression linear in leaf fields: {leaf: [Term]} with output block shape (nco, nzo)."""

    def __init__(self, nco, nzo, leaves=None):
        self.nco, self.nzo = nco, nzo
        self.leaves = leaves if leaves is not None else {}

    @classmethod
    def identity(cls, leaf, ncomp, nz):
        I = sparse.identity(nz, format="csr")
        return cls(ncomp, nz, {leaf: [Term(c, c, I) for c in range(ncomp)]})

    def map(self, fn, nco=None, nzo=None):
        """fn(term) -> iterable of new terms"""
        out = LinExpr(self.nco if nco is None else nco, self.nzo if nzo is None else nzo)
        for leaf, terms in self.leaves.items():
            new = []
            for t in terms:
                new.extend(fn(t))
            out.leaves[leaf] = new
        return out

    def scaled(self, c):
        return self.map(lambda t: [t.copy(coef=t.coef * c)])

    def added(self, other):
        assert (self.nco, self.nzo) == (other.nco, other.nzo), "LinExpr shape mismatch in add"
        out = LinExpr(self.nco, self.nzo, {k: list(v) for k, v in self.leaves.items()})
        for leaf, terms in other.leaves.items():
            out.leaves.setdefault(leaf, []).extend(terms)
        return out

    def apply_z(self, Zop):
        Zop = sparse.csr_matrix(Zop)
        return self.map(lambda t: [t.copy(Z=(Zop @ t.Z).tocsr())], nzo=Zop.shape[0])

    def fourier_diff(self, sep_index, order=1):
        def fn(t):
            n = t.copy(coef=t.coef * (1j) ** order)
            if sep_index == 0:
                n.ex = t.ex + order
            else:
                n.ey = t.ey + order
            return [n]
        return self.map(fn)

    def fourier_pin(self, sep_index, factor=1.0):
        """Restrict to the zero mode of a separable axis (constants, integrals, averages)."""
        def fn(t):
            n = t.copy(coef=t.coef * factor)
            if sep_index == 0:
                n.dx = 1
            else:
                n.dy = 1
            return [n]
        return self.map(fn)

    def comp_map(self, fn, nco):
        """fn(co) -> [(new_co, factor), ...]"""
        def tf(t):
            return [t.copy(co=c2, coef=t.coef * f) for (c2, f) in fn(t.co)]
        return self.map(tf, nco=nco)

    def with_dt(self):
        return self.map(lambda t: [t.copy(dt=t.dt + 1)])

    def split_dt(self):
        """(part without dt, part with exactly one dt)"""
        a = self.map(lambda t: [t] if t.dt == 0 else [])
        b = self.map(lambda t: [t.copy(dt=0)] if t.dt == 1 else [])
        for terms in self.leaves.values():
            if any(t.dt > 1 for t in terms):
                raise ValueError("second time derivatives are not supported")
        return a, b

    def is_empty(self):
        return all(len(v) == 0 for v in self.leaves.values())


def flatten(blocks, nrows, ncols, cutoff=1e-12):
    """blocks: iterable of (row0, nz_out, col0, nz_in, terms, force_dx, force_dy).
    Returns an element-level TermList (entry cutoff like core/subsystems.py:536)."""
    rows, cols, coefs, exs, eys, dxs, dys = [], [], [], [], [], [], []
    for (row0, nzo, col0, nzi, terms, fdx, fdy) in blocks:
        for t in terms:
            Z = t.Z.tocoo()
            if Z.nnz == 0:
                continue
            rows.append(row0 + t.co * nzo + Z.row)
            cols.append(col0 + t.ci * nzi + Z.col)
            coefs.append(t.coef * Z.data)
            n = Z.nnz
            exs.append(np.full(n, t.ex, np.int8))
            eys.append(np.full(n, t.ey, np.int8))
            dxs.append(np.full(n, 1 if (t.dx or fdx) else 0, np.int8))
            dys.append(np.full(n, 1 if (t.dy or fdy) else 0, np.int8))
    if not rows:
        return TermList(nrows, ncols)
    tl = TermList(nrows, ncols, np.concatenate(rows), np.concatenate(cols), np.concatenate(coefs),
                  np.concatenate(exs), np.concatenate(eys), np.concatenate(dxs), np.concatenate(dys))
    return tl.consolidated(cutoff)
