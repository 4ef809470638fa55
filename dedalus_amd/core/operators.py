"""
Operator / arithmetic expression nodes of the d3 API (Cartesian Fourier x Jacobi in this round).

Linear operators do not carry per-pencil matrices: `lin()` returns the operator's action as a
LinExpr (polynomial-in-wavenumber term list, core/polyop.py).  That one description serves both
the implicit matrices (the reference's expression_matrices recursion, core/operators.py:764-780,
925-946) and run-time evaluation (the reference's operate(), core/operators.py:984-995,
1628-1640, 2400-2413), which here is a single batched kernel launch over all pencils.

Nonlinear nodes (Multiply / DotProduct / CrossProduct / Power / UnaryGridFunction,
core/arithmetic.py:214-251, 666-674, 708-728, 855-866) are evaluated on the dealiased grid.
"""

import itertools
import numbers

import numpy as np
from scipy import sparse

from .domain import Domain
from .field import Field, Operand
from .polyop import LinExpr, Term


class NonlinearOperatorError(Exception):
    pass


def _cast(x, dist):
    if isinstance(x, Operand):
        return x
    if isinstance(x, numbers.Number):
        f = Field(dist, name=str(x))
        f["g"] = float(x)
        f._is_number = True
        f._number = float(x)
        return f
    raise TypeError("cannot use %r in a field expression" % (x,))


def _nonlinear_leaf(node, ctx, msg):
    """Outside the implicit (LHS) context a nonlinear node is a leaf of the linear expression above it."""
    if ctx is not None and not getattr(ctx, "strict", True):
        return LinExpr.identity(node, node.ncomp, node.dist.coupled_size(node.domain))
    raise NonlinearOperatorError(msg)


def _dist_of(*args):
    for a in args:
        if isinstance(a, Operand):
            return a.dist
    raise TypeError("no field operand")


class Future(Operand):
    """Base of all non-leaf nodes."""

    name = None

    def leaves(self):
        out = set()
        for a in self.args:
            if isinstance(a, Operand):
                out |= a.leaves()
        return out

    def has_dt(self):
        return any(isinstance(a, Operand) and a.has_dt() for a in self.args)

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join(repr(a) for a in self.args))


# ==================================================================================================
# helpers on domains / axes
# ==================================================================================================

def _sep_index(dist, axis):
    """Index (0 or 1) of a separable axis among the pencil cell axes."""
    return dist.separable_axes.index(axis)


def convert_linexpr(le, dist, dom_in, dom_out):
    """Convert coefficient data from dom_in to dom_out bases (Convert.operate, operators.py:1628-1640;
    ConvertJacobi basis.py:643-657; ConvertConstant* basis.py:660-676, 1180-1195)."""
    for ax, (bi, bo) in enumerate(zip(dom_in.by_axis, dom_out.by_axis)):
        if bi == bo:
            continue
        if bo is None:
            raise ValueError("cannot convert away a basis")
        if bo.separable:
            if bi is not None:
                raise ValueError("incompatible Fourier bases")
            le = le.fourier_pin(_sep_index(dist, ax), 1.0 / bo.constant_mode_value)
        else:
            if bi is None:
                col = sparse.csr_matrix(([1.0 / bo.constant_mode_value], ([0], [0])), shape=(bo.size, 1))
                le = le.apply_z(col)
            else:
                if bi.size != bo.size:
                    raise NotImplementedError("conversion between Jacobi bases of different size")
                le = le.apply_z(bi.convert_matrix(bo))
    return le


# ==================================================================================================
# linear nodes
# ==================================================================================================

class LinearOperator(Future):
    linear = True


class Convert(LinearOperator):
    def __init__(self, operand, domain):
        self.dist = operand.dist
        self.args = (operand,)
        self.operand = operand
        self.domain = domain
        self.tensorsig = operand.tensorsig

    def lin(self, ctx):
        return convert_linexpr(self.operand.lin(ctx), self.dist, self.operand.domain, self.domain)


class TimeDerivative(LinearOperator):
    def __init__(self, operand):
        self.dist, self.args, self.operand = operand.dist, (operand,), operand
        self.domain, self.tensorsig = operand.domain, operand.tensorsig

    def lin(self, ctx):
        return self.operand.lin(ctx).with_dt()

    def has_dt(self):
        return True


class Differentiate(LinearOperator):
    """d/dcoord (DifferentiateJacobi basis.py:679-697; DifferentiateRealFourier basis.py:1233-1260)."""

    def __init__(self, operand, coord):
        self.dist, self.args, self.operand, self.coord = operand.dist, (operand, coord), operand, coord
        self.axis = self.dist.coord_axis(coord)
        b = operand.domain.by_axis[self.axis]
        self.domain = operand.domain if b is None else operand.domain.replace(self.axis, b.derivative_basis(1))
        self.tensorsig = operand.tensorsig

    def lin(self, ctx):
        le = self.operand.lin(ctx)
        b = self.operand.domain.by_axis[self.axis]
        if b is None:
            return le.scaled(0.0)
        if b.separable:
            return le.fourier_diff(_sep_index(self.dist, self.axis))
        return le.apply_z(b.differentiate_matrix())


class Interpolate(LinearOperator):
    def __init__(self, operand, coord, position):
        self.dist, self.args, self.operand = operand.dist, (operand, coord, position), operand
        self.coord, self.position = coord, position
        self.axis = self.dist.coord_axis(coord)
        self.domain = operand.domain.replace(self.axis, None)
        self.tensorsig = operand.tensorsig

    def lin(self, ctx):
        le = self.operand.lin(ctx)
        b = self.operand.domain.by_axis[self.axis]
        if b is None:
            return le
        if b.separable:
            raise NotImplementedError("interpolation along a Fourier axis couples all modes: unsupported")
        pos = self.position
        if isinstance(pos, str):
            pos = {"left": b.bounds[0], "right": b.bounds[1], "center": 0.5 * (b.bounds[0] + b.bounds[1])}[pos]
        return le.apply_z(sparse.csr_matrix(b.interpolate_vector(pos)[None, :]))


class Integrate(LinearOperator):
    def __init__(self, operand, coords=None):
        self.dist, self.operand = operand.dist, operand
        if coords is None:
            axes = [ax for ax, b in enumerate(operand.domain.by_axis) if b is not None]
        else:
            if not isinstance(coords, (tuple, list)):
                coords = (coords,)
            axes = []
            for c in coords:
                for cc in getattr(c, "coords", (c,)):
                    axes.append(self.dist.coord_axis(cc))
        self.axes = axes
        self.args = (operand, coords)
        dom = operand.domain
        for ax in axes:
            dom = dom.replace(ax, None)
        self.domain, self.tensorsig = dom, operand.tensorsig
        self.average = False

    def lin(self, ctx):
        le = self.operand.lin(ctx)
        for ax in self.axes:
            b = self.operand.domain.by_axis[ax]
            if b is None:
                raise ValueError("cannot integrate along an axis without a basis")
            if b.separable:
                # integ cos(kx) dx = L delta_k0 (IntegrateRealFourier basis.py:1304-1325)
                le = le.fourier_pin(_sep_index(self.dist, ax), 1.0 if self.average else b.length)
            else:
                v = b.integrate_vector()
                if self.average:
                    v = v / b.length
                le = le.apply_z(sparse.csr_matrix(v[None, :]))
        return le


class Average(Integrate):
    def __init__(self, operand, coords=None):
        super().__init__(operand, coords)
        self.average = True


class Lift(LinearOperator):
    """Lift(operand, basis, n): operand * P_n(basis) (LiftJacobi basis.py:790-813)."""

    def __init__(self, operand, basis, n):
        self.dist, self.args, self.operand, self.basis, self.n = operand.dist, (operand, basis, n), operand, basis, n
        self.axis = self.dist.coord_axis(basis.coord)
        if operand.domain.by_axis[self.axis] is not None:
            raise ValueError("Lift operand must not have a basis along the lift axis")
        self.dist._register_domain(Domain(self.dist, (basis,)))
        self.domain = operand.domain.replace(self.axis, basis)
        self.tensorsig = operand.tensorsig

    def lin(self, ctx):
        n = self.n + self.basis.size if self.n < 0 else self.n
        col = sparse.csr_matrix(([1.0], ([n], [0])), shape=(self.basis.size, 1))
        return self.operand.lin(ctx).apply_z(col)


def _flat(idx, shape):
    out = 0
    for i, s in zip(idx, shape):
        out = out * s + i
    return out


def _unflat(c, shape):
    idx = []
    for s in reversed(shape):
        idx.append(c % s)
        c //= s
    return tuple(reversed(idx))


class Gradient(LinearOperator):
    """CartesianGradient (operators.py:2346-2413): out[i, ...] = d_i operand[...]."""

    def __init__(self, operand, coordsys=None):
        self.dist, self.operand = operand.dist, operand
        if coordsys is None:
            coordsys = self.dist.coordsystems[0]
        self.coordsys = coordsys
        self.args = (operand, coordsys)
        self.tensorsig = (coordsys,) + operand.tensorsig
        self.parts = [Differentiate(operand, c) for c in coordsys.coords]
        dom = self.parts[0].domain
        for p in self.parts[1:]:
            dom = dom.combine(p.domain, "add")
        self.domain = dom

    def lin(self, ctx):
        nc_in = self.operand.ncomp
        out = None
        for i, p in enumerate(self.parts):
            le = convert_linexpr(p.lin(ctx), self.dist, p.domain, self.domain)
            le = le.comp_map(lambda co, i=i: [(i * nc_in + co, 1.0)], nco=len(self.parts) * nc_in)
            out = le if out is None else out.added(le)
        return out


class Divergence(LinearOperator):
    """CartesianDivergence: contracts the first index, out[...] = d_i operand[i, ...]."""

    def __init__(self, operand, index=0):
        if index != 0 or not operand.tensorsig:
            raise NotImplementedError("divergence contracts the first tensor index")
        self.dist, self.operand, self.args = operand.dist, operand, (operand,)
        self.coordsys = operand.tensorsig[0]
        self.tensorsig = operand.tensorsig[1:]
        comps = [Component(operand, i) for i in range(self.coordsys.dim)]
        self.parts = [Differentiate(c, coord) for c, coord in zip(comps, self.coordsys.coords)]
        dom = self.parts[0].domain
        for p in self.parts[1:]:
            dom = dom.combine(p.domain, "add")
        self.domain = dom

    def lin(self, ctx):
        out = None
        for p in self.parts:
            le = convert_linexpr(p.lin(ctx), self.dist, p.domain, self.domain)
            out = le if out is None else out.added(le)
        return out


class Component(LinearOperator):
    """operand[i, ...] for the first tensor index."""

    def __init__(self, operand, i):
        self.dist, self.operand, self.i, self.args = operand.dist, operand, i, (operand, i)
        self.domain, self.tensorsig = operand.domain, operand.tensorsig[1:]

    def lin(self, ctx):
        rest = 1
        for cs in self.tensorsig:
            rest *= cs.dim
        i = self.i
        return self.operand.lin(ctx).comp_map(lambda co: [(co - i * rest, 1.0)] if co // rest == i else [], nco=rest)


def Laplacian(operand, coordsys=None):
    """CartesianLaplacian (operators.py:3322-3370) = div(grad())."""
    return Divergence(Gradient(operand, coordsys))


class Trace(LinearOperator):
    def __init__(self, operand):
        self.dist, self.operand, self.args = operand.dist, operand, (operand,)
        if len(operand.tensorsig) < 2 or operand.tensorsig[0].dim != operand.tensorsig[1].dim:
            raise ValueError("Trace needs a rank >= 2 tensor")
        self.domain, self.tensorsig = operand.domain, operand.tensorsig[2:]

    def lin(self, ctx):
        shape = self.operand.tshape
        rest = int(np.prod(shape[2:])) if len(shape) > 2 else 1

        def fn(co):
            idx = _unflat(co, shape)
            if idx[0] == idx[1]:
                return [(_flat(idx[2:], shape[2:]) if len(shape) > 2 else 0, 1.0)]
            return []
        return self.operand.lin(ctx).comp_map(fn, nco=rest)


class TransposeComponents(LinearOperator):
    def __init__(self, operand, indices=(0, 1)):
        self.dist, self.operand, self.args, self.indices = operand.dist, operand, (operand, indices), indices
        sig = list(operand.tensorsig)
        i, j = indices
        sig[i], sig[j] = sig[j], sig[i]
        self.domain, self.tensorsig = operand.domain, tuple(sig)

    def lin(self, ctx):
        shape_in, shape_out = self.operand.tshape, self.tshape
        i, j = self.indices

        def fn(co):
            idx = list(_unflat(co, shape_in))
            idx[i], idx[j] = idx[j], idx[i]
            return [(_flat(idx, shape_out), 1.0)]
        return self.operand.lin(ctx).comp_map(fn, nco=self.ncomp)


class Skew(LinearOperator):
    """2-D skew: (u_x, u_y) -> (-u_y, u_x) (operators.py Skew)."""

    def __init__(self, operand):
        self.dist, self.operand, self.args = operand.dist, operand, (operand,)
        if not operand.tensorsig or operand.tensorsig[0].dim != 2:
            raise ValueError("Skew needs a 2-D vector")
        self.domain, self.tensorsig = operand.domain, operand.tensorsig

    def lin(self, ctx):
        rest = self.ncomp // 2

        def fn(co):
            i, r = co // rest, co % rest
            return [(1 * rest + r, 1.0)] if i == 0 else [(0 * rest + r, -1.0)]
        return self.operand.lin(ctx).comp_map(fn, nco=self.ncomp)


# ==================================================================================================
# arithmetic
# ==================================================================================================

class Add(Future):
    def __init__(self, a, b):
        dist = _dist_of(a, b)
        a, b = _cast(a, dist), _cast(b, dist)
        if a.tensorsig != b.tensorsig:
            raise ValueError("Cannot add fields of different tensor signature")
        self.dist, self.args = dist, (a, b)
        self.domain = a.domain.combine(b.domain, "add")
        self.tensorsig = a.tensorsig

    def lin(self, ctx):
        a, b = self.args
        la = convert_linexpr(a.lin(ctx), self.dist, a.domain, self.domain)
        lb = convert_linexpr(b.lin(ctx), self.dist, b.domain, self.domain)
        return la.added(lb)


def _is_const_field(x, ctx):
    """A field without any basis that is not a problem variable acts as a constant coefficient."""
    return (isinstance(x, Field) and not x.domain.bases
            and (ctx is None or x not in getattr(ctx, "variables", ())))


class Multiply(Future):
    """a*b: tensor (outer) product, pointwise in space (arithmetic.py:MultiplyFields / MultiplyNumberField)."""

    def __init__(self, a, b):
        dist = _dist_of(a, b)
        self.dist = dist
        self.number = None
        if isinstance(a, numbers.Number) or isinstance(b, numbers.Number):
            if isinstance(b, numbers.Number):
                a, b = b, a
            if isinstance(b, numbers.Number):
                raise TypeError("use plain numbers")
            self.number = float(a)
            self.args = (a, b)
            self.domain, self.tensorsig = b.domain, b.tensorsig
            return
        self.args = (a, b)
        self.domain = a.domain.combine(b.domain, "mul")
        self.tensorsig = a.tensorsig + b.tensorsig

    def lin(self, ctx):
        a, b = self.args
        if self.number is not None:
            return b.lin(ctx).scaled(self.number)
        for (c, x, c_first) in ((a, b, True), (b, a, False)):
            if _is_const_field(c, ctx):
                vals = np.asarray(c["g"], dtype=float).reshape(-1)
                nc_c, nc_x = c.ncomp, x.ncomp
                le = x.lin(ctx)

                def fn(co, vals=vals, nc_c=nc_c, nc_x=nc_x, c_first=c_first):
                    out = []
                    for ic in range(nc_c):
                        if vals[ic] != 0.0:
                            out.append(((ic * nc_x + co) if c_first else (co * nc_c + ic), vals[ic]))
                    return out
                return le.comp_map(fn, nco=nc_c * nc_x)
        return _nonlinear_leaf(self, ctx, "product of two space-dependent fields is not linear")

    def bilinear_terms(self):
        a, b = self.args
        return [(ia * b.ncomp + ib, ia, ib, 1.0) for ia in range(a.ncomp) for ib in range(b.ncomp)]


class DotProduct(Future):
    """a@b: contracts the last index of a with the first of b (arithmetic.py:DotProduct :600-674)."""

    def __init__(self, a, b, indices=(-1, 0)):
        self.dist = a.dist
        self.args = (a, b)
        if not a.tensorsig or not b.tensorsig or a.tensorsig[-1].dim != b.tensorsig[0].dim:
            raise ValueError("DotProduct needs matching contracted indices")
        self.domain = a.domain.combine(b.domain, "mul")
        self.tensorsig = a.tensorsig[:-1] + b.tensorsig[1:]

    def lin(self, ctx):
        a, b = self.args
        for (c, x, c_first) in ((a, b, True), (b, a, False)):
            if _is_const_field(c, ctx):
                vals = np.asarray(c["g"], dtype=float).reshape(c.tshape)
                n = a.tensorsig[-1].dim
                le = x.lin(ctx)
                sa, sb = a.tshape, b.tshape
                shape_out = sa[:-1] + sb[1:]

                def fn(co, c_first=c_first):
                    out = []
                    idx = _unflat(co, x.tshape)
                    if c_first:      # c = a (constant), x = b: out[ia', ib'] += a[ia', k] b[k, ib']
                        k, ibr = idx[0], idx[1:]
                        for iar in itertools.product(*[range(s) for s in sa[:-1]]):
                            v = vals[iar + (k,)]
                            if v != 0.0:
                                out.append((_flat(iar + ibr, shape_out) if shape_out else 0, v))
                    else:            # c = b (constant), x = a
                        iar, k = idx[:-1], idx[-1]
                        for ibr in itertools.product(*[range(s) for s in sb[1:]]):
                            v = vals[(k,) + ibr]
                            if v != 0.0:
                                out.append((_flat(iar + ibr, shape_out) if shape_out else 0, v))
                    return out
                return le.comp_map(fn, nco=self.ncomp)
        return _nonlinear_leaf(self, ctx, "dot product of two space-dependent fields is not linear")

    def bilinear_terms(self):
        a, b = self.args
        sa, sb = a.tshape, b.tshape
        shape_out = sa[:-1] + sb[1:]
        terms = []
        for iar in itertools.product(*[range(s) for s in sa[:-1]]):
            for ibr in itertools.product(*[range(s) for s in sb[1:]]):
                ic = _flat(iar + ibr, shape_out) if shape_out else 0
                for k in range(sa[-1]):
                    terms.append((ic, _flat(iar + (k,), sa), _flat((k,) + ibr, sb), 1.0))
        return terms


class CrossProduct(Future):
    def __init__(self, a, b):
        self.dist, self.args = a.dist, (a, b)
        if a.tshape != (3,) or b.tshape != (3,):
            raise ValueError("CrossProduct needs two 3-D vectors")
        self.domain = a.domain.combine(b.domain, "mul")
        self.tensorsig = a.tensorsig

    def lin(self, ctx):
        return _nonlinear_leaf(self, ctx, "cross product is evaluated on the grid")

    def bilinear_terms(self):
        t = []
        for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
            t.append((i, j, k, 1.0))
            t.append((i, k, j, -1.0))
        return t


class Power(Future):
    def __init__(self, a, p):
        self.dist, self.args, self.p = a.dist, (a, p), p
        if a.tensorsig:
            raise ValueError("Power needs a scalar operand")
        self.domain = a.domain.combine(a.domain, "mul")
        self.tensorsig = ()

    def lin(self, ctx):
        return _nonlinear_leaf(self, ctx, "power is evaluated on the grid")


class UnaryGridFunction(Future):
    def __init__(self, func, a):
        self.dist, self.args, self.func = a.dist, (func, a), func
        self.domain = a.domain.combine(a.domain, "mul")
        self.tensorsig = a.tensorsig

    def leaves(self):
        return self.args[1].leaves()

    def has_dt(self):
        return False

    def lin(self, ctx):
        return _nonlinear_leaf(self, ctx, "grid functions are evaluated on the grid")


# ==================================================================================================
# d3-style function names
# ==================================================================================================

def grad(operand, coordsys=None):
    return Gradient(operand, coordsys)


def div(operand, index=0):
    return Divergence(operand, index)


def lap(operand, coordsys=None):
    return Laplacian(operand, coordsys)


def trace(operand):
    return Trace(operand)


def dot(a, b):
    return DotProduct(a, b)


def cross(a, b):
    return CrossProduct(a, b)


def skew(operand):
    return Skew(operand)


def transpose(operand, indices=(0, 1)):
    return TransposeComponents(operand, indices)


def integ(operand, *coords):
    return Integrate(operand, coords if coords else None)


def ave(operand, *coords):
    return Average(operand, coords if coords else None)


def interp(operand, **kw):
    return operand(**kw)


def dt(operand):
    return TimeDerivative(operand)


def lift(operand, basis, n):
    return Lift(operand, basis, n)


def curl(operand):
    """Cartesian 3-D curl: eps_ijk d_j u_k built from Differentiate/Component nodes."""
    cs = operand.tensorsig[0]
    if cs.dim != 3:
        raise NotImplementedError("curl is implemented for 3-D vectors")
    comps = [Component(operand, i) for i in range(3)]
    d = lambda j, k: Differentiate(comps[k], cs.coords[j])
    parts = [d(1, 2) - d(2, 1), d(2, 0) - d(0, 2), d(0, 1) - d(1, 0)]
    return StackComponents(parts, cs)


class StackComponents(LinearOperator):
    """Assemble a vector from scalar expressions (used by curl)."""

    def __init__(self, parts, coordsys):
        self.dist, self.args, self.parts = parts[0].dist, tuple(parts), parts
        dom = parts[0].domain
        for p in parts[1:]:
            dom = dom.combine(p.domain, "add")
        self.domain, self.tensorsig = dom, (coordsys,)

    def lin(self, ctx):
        out = None
        for i, p in enumerate(self.parts):
            le = convert_linexpr(p.lin(ctx), self.dist, p.domain, self.domain)
            le = le.comp_map(lambda co, i=i: [(i, 1.0)], nco=len(self.parts))
            out = le if out is None else out.added(le)
        return out
