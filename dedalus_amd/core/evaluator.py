"""
Expression evaluation on the device.

The reference oscillates every field through the layout chain and lets each operator `operate()`
wherever its layout condition is met (core/evaluator.py:95-146, core/future.py:149-206).  Here an
expression is split into alternating stages that each map to few large kernel launches:

  linear stage     every maximal sub-expression that is linear in its leaves is ONE batched
                   term-list mat-vec over all pencils (core/polyop.py); leaves that live in the same
                   system vector (the solver state) share a launch
  backward         coefficient -> dealiased grid, one fused FFT kernel per axis
  grid stage       products / dot / cross as a table-driven bilinear kernel
  forward          grid -> coefficient

Nothing leaves HBM between the stages.
"""

import os

import numpy as np

from . import operators as ops
from .field import Field
from .polyop import flatten
from .problems import LinCtx


class SystemBuffer:
    """A [rows][nx][ny] device array together with the leaves that are views into it.

    tiled = ny: the rows of `array` are stored tile-major ([kx / 8][ky / 8][kx % 8][ky % 8]; the solver's choice,
    SolverBase._enable_state_tiling): the kernels of the step read and write that layout, everything else -- user access to
    a state field, output, generic operators -- sees a NATURAL shadow of the rows it asks for, converted on demand
    (ddh_tile_rows) and valid until the solver writes the state again."""

    def __init__(self, array, nrows):
        self.array, self.nrows = array, nrows
        self.tiled = 0
        self.banded = 0             # rows of the whole vector when it is kx-band-major: [kx / 8][rows][ky / 8][8][8]
        self.ranges = []            # (row0, rows) of the variables
        self.valid = {}             # row0 -> the natural shadow of that variable's rows is current
        self._nat = None

    def _shadow(self, ex):
        if self._nat is None:
            self._nat = ex.empty(tuple(self.array.shape))
        return self._nat

    def invalidate(self):
        for k in self.valid:
            self.valid[k] = False

    def natural_rows(self, ex, row0, rows, current=True):
        """The natural shadow of rows row0 .. row0 + rows (a view); current: brought up to date from the tiled state."""
        nat = self._shadow(ex)
        if current and not self.valid.get(row0, False):
            _, nx, ny = self.array.shape
            ex.tile_rows(self.state_rows(row0, rows), nat[row0:row0 + rows], rows, nx, ny, False, self.banded)
            self.valid[row0] = True
        return nat[row0:row0 + rows]

    def state_rows(self, row0, rows):
        """Rows row0 .. row0 + rows of the state as it is stored: a view whose first element is the block's first (for a
        kx-band-major state a strided one, [rows][kx / 8][8 ny]: kernels take its address, nothing reshapes it)."""
        if not self.banded:
            return self.array[row0:row0 + rows]
        R, nx, ny = self.array.shape
        return self.array.as_strided((rows, nx // 8, 8 * ny), (8 * ny, R * 8 * ny, 1),
                                     self.array.storage_offset() + row0 * 8 * ny)

    def commit_rows(self, ex, row0, rows):
        """The natural shadow of these rows was written: into the tiled state with it."""
        _, nx, ny = self.array.shape
        ex.tile_rows(self._shadow(ex)[row0:row0 + rows], self.state_rows(row0, rows), rows, nx, ny, True, self.banded)
        self.valid[row0] = True

    def natural(self, ex):
        """The whole state in the natural layout (the array itself when it is not tiled)."""
        if not self.tiled:
            return self.array
        for row0, rows in self.ranges:
            self.natural_rows(ex, row0, rows)
        return self._nat


def _pencil_geom(dist):
    """(nf, nx, ny, kx, ky) of the separable axes shared by every system vector of this distributor."""
    return dist._pencil_geom


def set_pencil_geom(dist, domain_like_bases):
    sep = dist.separable_axes
    nf = len(sep)
    sizes, ks = [], []
    for ax in sep:
        b = domain_like_bases[ax]
        if b is None:
            raise ValueError("pencil geometry needs a basis on every separable axis")
        sizes.append(b.coeff_size)
        ks.append(b.mode_wavenumbers)
    nx = sizes[0] if nf >= 1 else 1
    ny = sizes[1] if nf >= 2 else 1
    kx = ks[0] if nf >= 1 else np.zeros(1)
    ky = ks[1] if nf >= 2 else np.zeros(1)
    mx_offset = 0
    if dist.size > 1:
        if nf < 1:
            raise NotImplementedError("several ranks need a separable axis to shard")
        lo, hi = dist.local_block(nx // 2)
        kx = kx[lo:hi]
        nx = 2 * (hi - lo)
        mx_offset = lo
    dist._pencil_geom = (nf, nx, ny, kx, ky)
    dist._mx_offset = mx_offset
    return dist._pencil_geom


def _full_sep(dist, domain):
    return all(domain.by_axis[ax] is not None for ax in dist.separable_axes)


class Evaluator:
    """Evaluates expressions for one distributor; caches term lists and pencil packs."""

    def __init__(self, dist, variables=()):
        self.dist = dist
        self.ex = dist.executor
        self.variables = tuple(variables)
        self.ctx = LinCtx(self.variables, strict=False)
        self._packs = {}
        self._lin_cache = {}
        self._const_cache = {}
        self._canon = {}
        self.cache = {}

    # ---- geometry ------------------------------------------------------------------------------------
    def geom(self):
        g = getattr(self.dist, "_pencil_geom", None)
        if g is None:
            # infer from any field with a full set of separable bases
            raise RuntimeError("pencil geometry not set (no field with all Fourier bases seen yet)")
        return g

    def pack(self, nrows_in):
        """Operator pack whose input vectors have `nrows_in` rows."""
        nf, nx, ny, kx, ky = self.geom()
        key = nrows_in
        if key not in self._packs:
            self._packs[key] = self.ex.make_pack(nf, nrows_in, nx, ny, kx, ky, self.dist._mx_offset)
        return self._packs[key]

    # ---- leaves ------------------------------------------------------------------------------------------
    def _leaf_rows(self, leaf):
        return leaf.ncomp * self.dist.coupled_size(leaf.domain)

    def _leaf_plane_data(self, leaf):
        """Coefficient data of a leaf as a [rows][nx][ny] system vector."""
        nf, nx, ny, kx, ky = self.geom()
        if isinstance(leaf, Field):
            c = leaf.coeff_data()
        else:
            c = self.eval_coeff(leaf)
        if _full_sep(self.dist, leaf.domain):
            return c
        # expand a field that lacks separable bases (constants, horizontally uniform fields)
        if isinstance(leaf, Field) and getattr(leaf, "_is_number", False):
            key = id(leaf)
            if key not in self._const_cache:
                self._const_cache[key] = self._expand(leaf, c)
            return self._const_cache[key]
        return self._expand(leaf, c)

    def _expand(self, leaf, c):
        nf, nx, ny, kx, ky = self.geom()
        rows = self._leaf_rows(leaf)
        host = self.ex.download(c).reshape(rows, -1)
        sep = self.dist.separable_axes
        sizes = [1 if leaf.domain.by_axis[ax] is None else leaf.domain.by_axis[ax].coeff_size for ax in sep]
        sx = sizes[0] if nf >= 1 else 1
        sy = sizes[1] if nf >= 2 else 1
        full = np.zeros((rows, nx, ny))
        if sx == nx or self.dist._mx_offset == 0:        # the kx = 0 pencil lives on the first rank
            full[:, :sx, :sy] = host.reshape(rows, sx, sy)
        return self.ex.from_host(full)

    # ---- linear stage ----------------------------------------------------------------------------------
    def _lin(self, expr):
        key = id(expr)
        if key not in self._lin_cache:
            self._lin_cache[key] = (expr, expr.lin(self.ctx))
        return self._lin_cache[key][1]

    def _force_flags(self, domain):
        sep = self.dist.separable_axes
        fx = len(sep) >= 1 and domain.by_axis[sep[0]] is None
        fy = len(sep) >= 2 and domain.by_axis[sep[1]] is None
        return fx, fy

    def apply_linear(self, le, out_domain, out_rows_total=None, row0=0, out=None, post_basis=None):
        """out[rows] = sum over leaves of TermList(le) @ leaf; returns a [rows][nx][ny] array.
        With post_basis (a Jacobi basis whose (a, b) differ from its grid parameters) the result is
        delivered in the grid basis instead: the ultraspherical back-substitution is fused into the
        mat-vec kernel (only possible when a single launch produces the result; else returns None)."""
        nf, nx, ny, kx, ky = self.geom()
        nrows = le.nco * le.nzo if out_rows_total is None else out_rows_total
        # group leaves by the system buffer they live in
        groups = {}
        for leaf, terms in le.leaves.items():
            if not terms:
                continue
            sb = getattr(leaf, "_sysbuf", None) if isinstance(leaf, Field) else None
            if sb is not None:
                groups.setdefault(("sys", id(sb)), [sb, []])[1].append((leaf, terms))
            else:
                groups.setdefault(("leaf", id(leaf)), [None, []])[1].append((leaf, terms))
        parts = []
        fdx_o, fdy_o = self._force_flags(out_domain)
        if post_basis is not None and len(groups) != 1:
            return None
        for key, (sb, items) in groups.items():
            ckey = (id(le), key, nrows, row0)
            cached = getattr(le, "_tl_cache", None)
            if cached is None:
                cached = le._tl_cache = {}
            if sb is not None:
                x = sb.natural(self.ex)         # (a tile-major state: its natural shadow)
                nrows_in = sb.nrows
            else:
                leaf = items[0][0]
                x = None
                nrows_in = self._leaf_rows(leaf)
            if ckey not in cached:
                blocks = []
                for leaf, terms in items:
                    nzi = self.dist.coupled_size(leaf.domain)
                    col0 = leaf._row0 if sb is not None else 0
                    fdx_i, fdy_i = self._force_flags(leaf.domain)
                    blocks.append((row0, le.nzo, col0, nzi, terms, fdx_o or fdx_i, fdy_o or fdy_i))
                tl = flatten(blocks, nrows, nrows_in)
                pack = self.pack(nrows_in)
                cached[ckey] = (pack, pack.add_matrix(tl))
            pack, mid = cached[ckey]
            if x is None:
                x = self._leaf_plane_data(items[0][0])
            y = self.ex.empty((nrows, nx, ny))
            if post_basis is not None:
                offs, bands = post_basis.conversion_bands()
                bkey = (post_basis.size, offs, bands.tobytes())
                ids = getattr(pack, "_bands_ids", None)
                if ids is None:
                    ids = pack._bands_ids = {}
                if bkey not in ids:
                    ids[bkey] = pack.add_upper_bands(post_basis.size, offs, bands)
                pack.matvec_solve(mid, ids[bkey], x, y)
            else:
                pack.matvec(mid, x, y)
            parts.append(y)
        if not parts:
            res = self.ex.zeros((nrows, nx, ny))
        elif len(parts) == 1:
            res = parts[0]
        else:
            res = self.ex.empty((nrows, nx, ny))
            self.ex.lincomb(res, parts, [1.0] * len(parts))
        return res

    # ---- structural de-duplication ------------------------------------------------------------------------
    def _skey(self, x):
        import numbers
        from .field import Operand
        if isinstance(x, Field):
            return ("F", id(x))
        if isinstance(x, Operand):
            k = getattr(x, "_skey_cache", None)
            if k is None:
                k = (type(x).__name__,) + tuple(self._skey(a) for a in x.args)
                x._skey_cache = k
            return k
        if isinstance(x, (numbers.Number, str, type(None))):
            return x
        if isinstance(x, (tuple, list)):
            return tuple(self._skey(a) for a in x)
        try:
            hash(x)
            return x
        except TypeError:
            return ("id", id(x))

    def canon(self, expr):
        """Structurally identical sub-expressions (e.g. `-u` written in two equations) share results."""
        if isinstance(expr, Field):
            return expr
        k = self._skey(expr)
        return self._canon.setdefault(k, expr)

    @staticmethod
    def _strip_scalar(expr):
        """(c, X) with expr == c * X: numeric factors are folded into the product kernels."""
        c = 1.0
        while isinstance(expr, ops.Multiply) and expr.number is not None:
            c *= expr.number
            expr = expr.args[1]
        return c, expr

    # ---- public evaluation -------------------------------------------------------------------------------
    def eval_coeff(self, expr):
        """Coefficient data [ncomp, storage coeff shape] of expr in expr.domain."""
        expr = self.canon(expr)
        key = ("c", id(expr))
        if key in self.cache:
            return self.cache[key]
        if isinstance(expr, Field):
            res = expr.coeff_data()
        elif self._is_nonlinear_node(expr):
            g = self.eval_grid(expr)
            res = self.ex.empty((expr.ncomp,) + tuple(expr.domain.storage_coeff_shape()))
            self.dist.transformer.forward_data(expr.domain, expr.ncomp, g, expr.domain.dealias, res)
        else:
            le = self._lin(expr)
            if not _full_sep(self.dist, expr.domain):
                raise NotImplementedError("evaluating expressions without all Fourier bases")
            res = self.apply_linear(le, expr.domain)
            res = res.reshape((expr.ncomp,) + tuple(expr.domain.storage_coeff_shape()))
        self.cache[key] = res
        return res

    def eval_grid(self, expr):
        """Grid data at the dealias scales."""
        expr = self.canon(expr)
        key = ("g", id(expr))
        if key in self.cache:
            return self.cache[key]
        scales = expr.domain.dealias
        if isinstance(expr, Field):
            res = expr.grid_data(scales)
        elif self._is_nonlinear_node(expr):
            res = self._eval_nonlinear(expr, scales)
        else:
            res = self._linear_to_grid_fused(expr, scales)
            if res is None:
                c = self.eval_coeff(expr)
                res = self.ex.empty((expr.ncomp,) + tuple(expr.domain.storage_grid_shape(scales)))
                self.dist.transformer.backward_data(expr.domain, expr.ncomp, c, res, scales)
        self.cache[key] = res
        return res

    def eval_stage(self, leaf, k):
        """Data of a Field after the first k backward transform steps (k = 0: coefficients)."""
        key = ("s", id(leaf), k)
        if key in self.cache:
            return self.cache[key]
        dom = leaf.domain
        if k == 0:
            res = leaf.coeff_data().reshape((leaf.ncomp,) + tuple(dom.storage_coeff_shape()))
        else:
            prev = self.eval_stage(leaf, k - 1) if k > 1 else None          # (k == 1: _stage0, below / in _z_dual)
            res = None
            if k == 1 and self.dist.size > 1:
                order = self.__dict__.setdefault("_stage1_order", [])
                if all(x is not leaf for x in order):
                    order.append(leaf)              # prefetch_stage1
            wz = getattr(self, "_dualz_want", {}).get(id(leaf)) if k == 1 else None
            if wz is not None:
                self._z_dual(leaf, *wz)                 # the z-derivative is an operand too: both from one read
                res = self.cache.get(key)
            want = getattr(self, "_dual_want", {}).get((id(leaf), k - 1))
            if res is None and want is not None and os.environ.get("DDH_NO_DUAL_FFT") is None:
                # this field AND its derivative along the axis of step k - 1 are operands: one kernel, one read
                if prev is None:
                    prev = self.eval_stage(leaf, 0)
                pair = self.dist.transformer.backward_dual_step(dom, leaf.ncomp, prev, dom.dealias, k - 1, want)
                if pair is not None:
                    res, self.cache[("sd", id(leaf), k - 1, want)] = pair
            if res is None:
                ctile = 0
                if k == 1:
                    prev, ctile = self._stage0(leaf)
                res = self.dist.transformer.backward_steps(dom, leaf.ncomp, prev, dom.dealias, k - 1, k,
                                                           **(dict(ctile=ctile) if ctile else {}))
        self.cache[key] = res
        return res

    def _stage0(self, leaf):
        """(coefficient data of a field for its first backward step, row length of tile-major rows or 0): a state field
        kept tile-major by its solver is handed to the z transform as it lies when that transform reads the layout
        (Transformer.coeff_tiled_ok), otherwise its natural data."""
        tr = self.dist.transformer
        if getattr(leaf, "_tiled", None) is not None and hasattr(tr, "coeff_tiled_ok") \
                and tr.coeff_tiled_ok(leaf.domain, leaf.domain.dealias, int(leaf._tiled.tiled)):
            c, ctile = leaf.coeff_tiled()              # (handed on as it lies: the transforms take its address)
            return c, ((ctile, int(leaf._tiled.banded)) if leaf._tiled.banded else ctile)
        return self.eval_stage(leaf, 0), 0

    def eval_pregrid(self, expr):
        """Data of a field / linear expression with every axis but the last storage axis in (dealiased)
        grid space: the operand layout of the fused grid stage."""
        expr = self.canon(expr)
        key = ("p", id(expr))
        if key in self.cache:
            return self.cache[key]
        scales = expr.domain.dealias
        tr = self.dist.transformer
        if self._is_nonlinear_node(expr):
            raise NotImplementedError("pre-grid data of a nonlinear node")
        if isinstance(expr, Field):
            res = self.eval_stage(expr, max(tr.nsteps(expr.domain, scales) - 1, 0))
        else:
            res = self._le_to_grid(self._lin(expr), expr.domain, expr.ncomp, scales, skip_last=True)
        self.cache[key] = res
        return res

    # ---- operands of the fused grid stage -------------------------------------------------------------------
    def _derived_component(self, x, items, steps):
        """(leaf, ci, step, dscale) when component `items` of the linear expression x is a Fourier
        derivative of one component of a field (up to the Jacobi conversion that the backward transform
        undoes): its grid data then comes from the field's own partially transformed data, with the
        derivative applied while that axis' coefficients are loaded."""
        from .polyop import LinExpr
        if len(items) != 1:
            return None
        leaf, t = items[0]
        if not isinstance(leaf, Field) or getattr(leaf, "_is_number", False):
            return None
        if t.dx or t.dy or t.dt or (t.ex + t.ey) != 1:
            return None
        ld, xd = leaf.domain, x.domain
        if ld.dealias != xd.dealias or ld.storage_coeff_shape() != xd.storage_coeff_shape():
            return None
        if ld.storage_grid_shape(ld.dealias) != xd.storage_grid_shape(xd.dealias):
            return None
        try:
            ident = LinExpr.identity(leaf, 1, self.dist.coupled_size(ld))
            expect = ops.convert_linexpr(ident, self.dist, ld, xd).leaves[leaf][0]
        except (ValueError, NotImplementedError):
            return None
        if expect.dx or expect.dy or expect.Z.shape != t.Z.shape:
            return None
        diff = (expect.Z - t.Z)
        if diff.nnz and abs(diff).max() > 1e-13:
            return None
        sep = self.dist.separable_axes
        axis = sep[0] if t.ex else sep[1]
        basis = xd.by_axis[axis]
        r = t.coef / (1j * expect.coef)
        if basis is None or abs(r.imag) > 1e-14 * max(abs(r), 1.0):
            return None
        step = [i for i, (pos, b, spec) in enumerate(steps) if self.dist.storage_order[pos] == axis]
        if len(step) != 1 or steps[step[0]][2][0] != "rfft":
            return None
        return (leaf, t.ci, step[0], float(r.real) * 2.0 * np.pi / basis.length)

    def _z_derived_component(self, x, items, steps):
        """(leaf, ci, dvec) when component `items` of x is a one-superdiagonal operator along the Jacobi axis -- the
        z-derivative (DifferentiateJacobi, core/basis.py:806-840: T_n -> the (a0+1, b0+1) family, one superdiagonal) -- of
        one component of a field kept in the family's own basis: its grid data then comes out of the field's own backward
        z transform, second pass (Transformer.backward_dual_z), instead of a sparse mat-vec and a transform of its own."""
        if len(items) != 1 or not hasattr(self.dist.executor, "transform_dual_z"):
            return None
        leaf, t = items[0]
        if not isinstance(leaf, Field) or getattr(leaf, "_is_number", False):
            return None
        if t.ex or t.ey or t.dx or t.dy or t.dt or t.coef.imag != 0.0:
            return None
        ld, xd = leaf.domain, x.domain
        if ld.dealias != xd.dealias or ld.storage_coeff_shape() != xd.storage_coeff_shape():
            return None
        if ld.storage_grid_shape(ld.dealias) != xd.storage_grid_shape(xd.dealias):
            return None
        lsteps = self.dist.transformer._steps(ld, ld.dealias)
        if not steps or not lsteps or steps[0][2][0] != "cheb" or lsteps[0][2][0] != "cheb":
            return None
        if steps[0][0] != 0 or lsteps[0][0] != 0 or tuple(steps[0][2][1:3]) != tuple(lsteps[0][2][1:3]):
            return None
        bl, bx = lsteps[0][1], steps[0][1]
        if (bl.a, bl.b) != (bl.a0, bl.b0) or (bx.a, bx.b) == (bx.a0, bx.b0):
            return None                                 # pass 0 has no conversion solve, pass 1 needs the plan's
        Z = t.Z.tocoo()
        M = steps[0][2][2]
        if Z.shape != (M, M) or Z.nnz == 0 or np.any(Z.col - Z.row != 1):
            return None
        dvec = np.zeros(M)
        np.add.at(dvec, Z.row, Z.data.real * t.coef.real)
        if np.iscomplexobj(Z.data) and np.abs(Z.data.imag).max() > 0:
            return None
        return (leaf, t.ci, dvec)

    def _z_dual(self, leaf, xdomain, dvec):
        """Stage-1 data (z in grid space) of the z-derivative described by dvec, produced together with the field's own
        stage-1 data (cache key ("s", id(leaf), 1))."""
        key = ("sz", id(leaf), dvec.tobytes())
        if key in self.cache:
            return self.cache[key]
        store = self.__dict__.setdefault("_dvec_dev", {})
        dv = store.get(dvec.tobytes())
        if dv is None:
            dv = store[dvec.tobytes()] = self.dist.executor.from_host(dvec)
        if self.dist.size > 1:
            order = self.__dict__.setdefault("_stage1_order", [])
            if all(x is not leaf for x in order):
                order.append(leaf)                  # prefetch_stage1 (reached here before eval_stage(leaf, 1) ever ran)
        coeff, ctile = self._stage0(leaf)
        plain, der = self.dist.transformer.backward_dual_z(leaf.domain, xdomain, leaf.ncomp, coeff, leaf.domain.dealias, dv,
                                                           **(dict(ctile=ctile) if ctile else {}))
        self.cache.setdefault(("s", id(leaf), 1), plain)
        self.cache[key] = der
        return der

    def _component_plan(self, x):
        """Split the components of a linear expression into Fourier-derivative components (see
        _derived_component) and generic ones (evaluated by the batched mat-vec)."""
        plan = getattr(x, "_fuse_plan", None)
        if plan is not None:
            return plan
        from .polyop import LinExpr
        le = self._lin(x)
        steps = self.dist.transformer._steps(x.domain, x.domain.dealias)
        per = {co: [] for co in range(x.ncomp)}
        for leaf, terms in le.leaves.items():
            for t in terms:
                per[t.co].append((leaf, t))
        derived, generic = {}, []
        no_deriv = os.environ.get("DDH_NO_DERIV_AT_LOAD") is not None
        zder = {}
        no_zdual = os.environ.get("DDH_NO_DUAL_Z") is not None
        for co in range(x.ncomp):
            d = None if no_deriv else self._derived_component(x, per[co], steps)
            if d is None:
                z = None if no_zdual else self._z_derived_component(x, per[co], steps)
                if z is None:
                    generic.append(co)
                else:
                    zder[co] = z
            else:
                derived[co] = d
        x._fuse_zder = zder
        sub = None
        if generic:
            idx = {co: i for i, co in enumerate(generic)}
            sub = LinExpr(len(generic), le.nzo,
                          {leaf: [t.copy(co=idx[t.co]) for t in terms if t.co in idx] for leaf, terms in le.leaves.items()})
        plan = x._fuse_plan = (derived, generic, sub)
        return plan

    def _operand_lines(self, x):
        """Per component of a product factor: (parent array, index, dscale) such that parent[index] holds the
        component's pre-grid lines and dscale != 0 asks the fused kernel to differentiate at load."""
        x = self.canon(x)
        key = ("ol", id(x))
        if key in self.cache:
            return self.cache[key]
        if isinstance(x, Field):
            pg = self.eval_pregrid(x)
            res = [(pg, i, 0.0) for i in range(x.ncomp)]
            self.cache[key] = res
            return res
        derived, generic, sub = self._component_plan(x)
        tr = self.dist.transformer
        scales = x.domain.dealias
        res = [None] * x.ncomp
        if generic:
            arr = self._le_to_grid(sub, x.domain, len(generic), scales, skip_last=True)
            for i, co in enumerate(generic):
                res[co] = (arr, i, 0.0)
        zgroups = {}
        for co, (leaf, ci, dvec) in getattr(x, "_fuse_zder", {}).items():
            zgroups.setdefault((id(leaf), dvec.tobytes()), [leaf, dvec, []])[2].append((co, ci))
        for leaf, dvec, members in zgroups.values():
            der = self._z_dual(leaf, x.domain, dvec)
            n = tr.nsteps(leaf.domain, scales)
            out = tr.backward_steps(leaf.domain, leaf.ncomp, der, scales, 1, n - 1) if n - 1 > 1 else der
            for co, ci in members:
                res[co] = (out, ci, 0.0)
        groups = {}
        for co, (leaf, ci, step, dscale) in derived.items():
            groups.setdefault((id(leaf), step, dscale), [leaf, []])[1].append((co, ci))
        for (lid, step, dscale), (leaf, members) in groups.items():
            n = tr.nsteps(leaf.domain, scales)
            if step == n - 1:
                pg = self.eval_stage(leaf, n - 1)
                for co, ci in members:
                    res[co] = (pg, ci, dscale)
                continue
            stash = None
            if step + 1 == n - 1:
                if getattr(self, "_dual_want", {}).get((lid, step)) == dscale and ("s", lid, step + 1) not in self.cache:
                    self.eval_stage(leaf, step + 1)     # the field's own transform is needed too: make both now
                stash = self.cache.get(("sd", lid, step, dscale))
            if stash is not None:                       # produced together with the field's own transform
                for co, ci in members:
                    res[co] = (stash, ci, 0.0)
                continue
            src = self.eval_stage(leaf, step)
            cis = sorted({ci for _, ci in members})
            if cis == list(range(leaf.ncomp)):
                out = tr.backward_steps(leaf.domain, leaf.ncomp, src, scales, step, n - 1, deriv=(step, dscale))
                for co, ci in members:
                    res[co] = (out, ci, 0.0)
            else:
                outs = {ci: tr.backward_steps(leaf.domain, 1, src[ci:ci + 1], scales, step, n - 1,
                                              deriv=(step, dscale)) for ci in cis}
                for co, ci in members:
                    res[co] = (outs[ci], 0, 0.0)
        self.cache[key] = res
        return res

    def fusable_product(self, expr):
        """(a, b, terms, basis, spec) when the product node can run in the fused grid stage: both factors
        linear in fields, on the product's full domain, last storage axis a contiguous RealFourier axis.
        `a` is the factor with fewer components (kept in registers by the kernel)."""
        expr = self.canon(expr)
        if not isinstance(expr, (ops.Multiply, ops.DotProduct, ops.CrossProduct)) or not self._is_nonlinear_node(expr):
            return None
        last = self.dist.transformer.fusable_last_axis(expr.domain, expr.domain.dealias)
        if last is None:
            return None
        a, b = expr.args
        sa, a0 = self._strip_scalar(a)
        sb, b0 = self._strip_scalar(b)
        a0, b0 = self.canon(a0), self.canon(b0)
        for x in (a0, b0):
            if self._is_nonlinear_node(x) or x.domain.dealias != expr.domain.dealias:
                return None
            if (x.domain.storage_coeff_shape() != expr.domain.storage_coeff_shape()
                    or x.domain.storage_grid_shape(x.domain.dealias) != expr.domain.storage_grid_shape(x.domain.dealias)):
                return None                     # broadcasting products stay on the unfused path
            if not isinstance(x, Field) and any(self._is_nonlinear_node(l) for l in self._lin(x).leaves):
                return None
        terms = [(ic, ia, ib, cf * sa * sb) for (ic, ia, ib, cf) in expr.bilinear_terms()]
        if b0.ncomp < a0.ncomp:
            a0, b0 = b0, a0
            terms = [(ic, ib, ia, cf) for (ic, ia, ib, cf) in terms]
        lim = self.ex.FUSED_LIMITS
        if a0.ncomp > lim["na"] or b0.ncomp > lim["nb"] or expr.ncomp > lim["nc"] or len(terms) > lim["terms"]:
            return None
        return a0, b0, terms, last[0], last[1]

    def _plan_dual_transforms(self, operands):
        """Fields whose transform along a (non-last) RealFourier step is needed both plain and differentiated -- e.g. u and
        d/dx u of u.grad(u): eval_stage then produces both from one read (Transformer.backward_dual_step).  The plain
        transform counts as needed when the field itself or its last-axis derivative (taken from the field's own
        pre-grid lines) is an operand."""
        want = getattr(self, "_dual_want", None)
        if want is None:
            want = self._dual_want = {}
        done = self.__dict__.setdefault("_dual_planned", set())
        gkey = tuple(id(x) for x in operands)
        if gkey in done:                 # the expression graph is static: planned once per group of operands
            return
        done.add(gkey)
        tr = self.dist.transformer
        plain, derived = set(), {}
        for x in operands:
            x = self.canon(x)
            if isinstance(x, Field):
                plain.add(id(x))
                continue
            d, generic, sub = self._component_plan(x)
            zw = self.__dict__.setdefault("_dualz_want", {})
            for co, (leaf, ci, dvec) in getattr(x, "_fuse_zder", {}).items():
                zw.setdefault(id(leaf), (x.domain, dvec))
            n = None
            per = {}
            for co, (leaf, ci, step, dscale) in d.items():
                n = tr.nsteps(leaf.domain, leaf.domain.dealias)
                if step == n - 1:
                    plain.add(id(leaf))
                else:
                    per.setdefault((id(leaf), step, dscale), [leaf, set()])[1].add(ci)
            for (lid, step, dscale), (leaf, cis) in per.items():
                if cis == set(range(leaf.ncomp)) and step + 1 == tr.nsteps(leaf.domain, leaf.domain.dealias) - 1:
                    derived[(lid, step)] = dscale
        for (lid, step), dscale in derived.items():
            if lid in plain:
                want[(lid, step)] = dscale

    def eval_fused_products(self, group, outs, scales=None):
        """One fused launch for product nodes that share their register operand `a`.
        group: [(expr, (a, b, terms, basis, spec))]; outs: pre-grid result arrays [ncomp, ..., M]; scales: a factor per
        product folded into its term coefficients (the sign / constant of the right-hand side it feeds)."""
        a = group[0][1][0]
        basis, spec = group[0][1][3], group[0][1][4]
        self._plan_dual_transforms([a] + [fp[1] for _, fp in group])
        la = self._operand_lines(a)
        a_list = [par[i] for (par, i, ds) in la]
        a_ds = [ds for (par, i, ds) in la]
        b_list, b_ds, out_list, terms, bpos = [], [], [], [], {}
        for gi, ((expr, (a_, b, tms, _, _)), out) in enumerate(zip(group, outs)):
            sc = 1.0 if scales is None else float(scales[gi])
            lb = self._operand_lines(b)
            ob = len(out_list)
            out_list.extend(out[i] for i in range(expr.ncomp))
            for (ic, ia, ib, cf) in tms:
                par, i, ds = lb[ib]
                kb = (id(par), i, ds)
                if kb not in bpos:
                    bpos[kb] = len(b_list)
                    b_list.append(par[i])
                    b_ds.append(ds)
                terms.append((ob + ic, ia, bpos[kb], cf * sc))
        if len(b_list) > self.ex.FUSED_LIMITS["nb"]:
            raise RuntimeError("fused grid stage: too many distinct operands")
        M = spec[2]
        nlines = int(np.prod(a_list[0].shape)) // M
        self.ex.rfft_bilinear_fused(spec, basis, a_list, b_list, out_list, nlines, terms, a_ds, b_ds)

    def _le_to_grid(self, le, domain, ncomp, scales, skip_last=False):
        """Grid (or pre-grid) data of a linear expression given as a LinExpr on `domain`."""
        tr = self.dist.transformer
        if skip_last:
            self._generic_pregrid = True            # (a whole-array operand of the grid stage: no windows, windows_possible)
            if getattr(tr, "_win", None) is not None and tr._win["k"] is not None:
                raise RuntimeError("windowed evaluation met a generic pre-grid operand")
        if not _full_sep(self.dist, domain):
            raise NotImplementedError("evaluating expressions without all Fourier bases")
        jac = [ax for ax in self.dist._jacobi_axes if domain.by_axis[ax] is not None]
        if len(jac) == 1:
            b = domain.by_axis[jac[0]]
            if (b.a, b.b) != (b.a0, b.b0) and b.a0 == b.b0 == -0.5:
                c = self.apply_linear(le, domain, post_basis=b)
                if c is not None:
                    gdom = domain.replace(jac[0], b.clone_with(a=b.a0, b=b.b0))
                    c = c.reshape((ncomp,) + tuple(gdom.storage_coeff_shape()))
                    shape = tr.pregrid_shape(gdom, ncomp, scales) if skip_last else \
                        (ncomp,) + tuple(gdom.storage_grid_shape(scales))
                    res = self.ex.empty(shape)
                    tr.backward_data(gdom, ncomp, c, res, scales, skip_last=skip_last)
                    return res
        c = self.apply_linear(le, domain).reshape((ncomp,) + tuple(domain.storage_coeff_shape()))
        shape = tr.pregrid_shape(domain, ncomp, scales) if skip_last else (ncomp,) + tuple(domain.storage_grid_shape(scales))
        res = self.ex.empty(shape)
        tr.backward_data(domain, ncomp, c, res, scales, skip_last=skip_last)
        return res

    def _linear_to_grid_fused(self, expr, scales, skip_last=False):
        """Linear expression -> grid with the ultraspherical conversion solve fused into the mat-vec."""
        if ("c", id(expr)) in self.cache or not _full_sep(self.dist, expr.domain):
            return None
        jac = [ax for ax in self.dist._jacobi_axes if expr.domain.by_axis[ax] is not None]
        if len(jac) != 1:
            return None
        b = expr.domain.by_axis[jac[0]]
        if (b.a, b.b) == (b.a0, b.b0) or not (b.a0 == b.b0 == -0.5):
            return None
        le = self._lin(expr)
        c = self.apply_linear(le, expr.domain, post_basis=b)
        if c is None:
            return None
        gdom = expr.domain.replace(jac[0], b.clone_with(a=b.a0, b=b.b0))
        c = c.reshape((expr.ncomp,) + tuple(gdom.storage_coeff_shape()))
        tr = self.dist.transformer
        if skip_last:
            res = self.ex.empty(tr.pregrid_shape(gdom, expr.ncomp, scales))
        else:
            res = self.ex.empty((expr.ncomp,) + tuple(gdom.storage_grid_shape(scales)))
        tr.backward_data(gdom, expr.ncomp, c, res, scales, skip_last=skip_last)
        return res

    @staticmethod
    def _is_nonlinear_node(expr):
        if isinstance(expr, (ops.CrossProduct, ops.Power, ops.UnaryGridFunction)):
            return True
        if isinstance(expr, (ops.Multiply, ops.DotProduct)):
            if isinstance(expr, ops.Multiply) and expr.number is not None:
                return False
            a, b = expr.args
            return not (ops._is_const_field(a, None) or ops._is_const_field(b, None))
        return False

    def _grid_shape(self, expr, scales):
        return (expr.ncomp,) + tuple(expr.domain.storage_grid_shape(scales))

    def _broadcast_grid(self, operand, out_domain, scales):
        """Grid data of an operand broadcast to the product's full grid shape."""
        g = self.eval_grid(operand)
        tgt = tuple(out_domain.storage_grid_shape(scales))
        src = tuple(operand.domain.storage_grid_shape(scales))
        if src == tgt:
            return g
        host = self.ex.download(g).reshape((operand.ncomp,) + src)
        full = np.broadcast_to(host, (operand.ncomp,) + tgt)
        return self.ex.from_host(np.ascontiguousarray(full))

    def _eval_nonlinear(self, expr, scales):
        ex = self.ex
        if isinstance(expr, (ops.Multiply, ops.DotProduct, ops.CrossProduct)):
            a, b = expr.args
            sa, a0 = self._strip_scalar(a)
            sb, b0 = self._strip_scalar(b)
            ga = self._broadcast_grid(a0, expr.domain, scales)
            gb = self._broadcast_grid(b0, expr.domain, scales)
            out = ex.empty(self._grid_shape(expr, scales))
            npts = int(np.prod(expr.domain.storage_grid_shape(scales)))
            terms = expr.bilinear_terms()
            if sa * sb != 1.0:
                terms = [(ic, ia, ib, cf * sa * sb) for (ic, ia, ib, cf) in terms]
            ex.bilinear(out, expr.ncomp, ga, gb, npts, terms)
            return out
        if isinstance(expr, ops.Power):
            a, p = expr.args
            ga = self._broadcast_grid(a, expr.domain, scales)
            if p == 2:
                out = ex.empty(self._grid_shape(expr, scales))
                npts = int(np.prod(expr.domain.storage_grid_shape(scales)))
                ex.bilinear(out, 1, ga, ga, npts, [(0, 0, 0, 1.0)])
                return out
            return ex.from_host(self.ex.download(ga) ** p)      # analysis-only path
        if isinstance(expr, ops.UnaryGridFunction):
            func, a = expr.args
            ga = self.eval_grid(a)
            return ex.from_host(func(self.ex.download(ga)))     # analysis-only path
        raise NotImplementedError(type(expr))

    def new_pass(self):
        self.cache = {}
        settle = getattr(self.dist.transformer, "settle", None)
        if settle is not None:
            settle()                            # no exchange outlives the pass that started it

    def window_cache_reset(self):
        """Between two windows of z planes (Transformer windows): everything computed from a window is dropped, the fields'
        coefficient data and their z-step results (whole arrays, exchanged in windows) stay."""
        self.cache = {k: v for k, v in self.cache.items()
                      if (k[0] == "s" and k[2] <= 1) or k[0] == "sz"}

    def windows_possible(self):
        """The previous pass evaluated every pre-grid operand through the stage cache (eval_stage / _z_dual): only then
        can the grid stage run in windows (a generic operand is a whole array)."""
        return bool(self.__dict__.get("_stage1_order")) and not self.__dict__.get("_generic_pregrid", False)

    def prefetch_stage1(self, force=False):
        """Several ranks: issue the z step (and with it the pencil transposes, which return without waiting:
        Transformer._defer) of every field the previous pass transformed, in that pass's order, before anything consumes
        one -- the exchanges queue up on the communicator's stream and the wire of a later field runs under the x transforms
        of an earlier one.  The expression graph is static, so the previous pass's list is this pass's list; a field that
        turns out not to be needed costs one transform, a new one is simply evaluated on demand.  DDH_A2A_PREFETCH=1 switches
        it on: under an emulated wire it measured neutral (the grid stage needs every operand, so only the x transforms
        are there to hide the wire behind: profiles/r6_rank_emulation.txt) and it is off by default."""
        order = self.__dict__.get("_stage1_order")
        if not order or self.dist.size == 1 or (not force and os.environ.get("DDH_A2A_PREFETCH", "0") != "1"):
            return
        for leaf in order:
            self.eval_stage(leaf, 1)


def evaluate_expression(expr):
    """expr.evaluate(): returns a new Field holding the value (operators.py Future.evaluate)."""
    dist = expr.dist
    ev = getattr(dist, "_default_evaluator", None)
    if ev is None:
        ev = dist._default_evaluator = Evaluator(dist)
    ev.new_pass()
    out = Field(dist, bases=expr.domain.bases, tensorsig=expr.tensorsig)
    if ev._is_nonlinear_node(expr):
        g = ev.eval_grid(expr)
        out._g, out._g_scales = g, expr.domain.dealias
        out.scales = expr.domain.dealias
        out.layout = "g"
    else:
        c = ev.eval_coeff(expr)
        out._c = c.reshape(out._storage_shape("c", None))
        out.layout = "c"
    out._authority = "device"
    ev.new_pass()
    return out
