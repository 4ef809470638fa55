"""
Operands and Fields.

A Field's authoritative data lives in HBM (coefficient array and/or grid array, "z-major" storage
order: Jacobi axis outermost, Fourier axes inner).  `field['g']` /
`field['c']` hand out a persistent host mirror in the reference's axis order
(core/field.py:578-597) and mark the host as authoritative until the next device use -- these are
the host<->device synchronisation points (seam B5).
"""

import numbers

import numpy as np

from .domain import Domain


class Operand:
    """Arithmetic front-end shared by Fields and operator nodes (core/field.py:40-342 role)."""

    __array_priority__ = 100.0

    def __add__(self, other):
        from . import operators as ops
        return ops.Add(self, other)

    def __radd__(self, other):
        from . import operators as ops
        return ops.Add(other, self)

    def __sub__(self, other):
        from . import operators as ops
        if isinstance(other, numbers.Number):
            return ops.Add(self, -other)
        return ops.Add(self, ops.Multiply(-1, other))

    def __rsub__(self, other):
        from . import operators as ops
        return ops.Add(other, ops.Multiply(-1, self))

    def __neg__(self):
        from . import operators as ops
        return ops.Multiply(-1, self)

    def __pos__(self):
        return self

    def __mul__(self, other):
        from . import operators as ops
        return ops.Multiply(self, other)

    def __rmul__(self, other):
        from . import operators as ops
        return ops.Multiply(other, self)

    def __truediv__(self, other):
        from . import operators as ops
        if isinstance(other, numbers.Number):
            return ops.Multiply(1.0 / other, self)
        return ops.Multiply(self, ops.Power(other, -1))

    def __rtruediv__(self, other):
        from . import operators as ops
        return ops.Multiply(other, ops.Power(self, -1))

    def __matmul__(self, other):
        from . import operators as ops
        return ops.DotProduct(self, other)

    def __rmatmul__(self, other):
        from . import operators as ops
        return ops.DotProduct(other, self)

    def __pow__(self, p):
        from . import operators as ops
        return ops.Power(self, p)

    def __call__(self, **kw):
        """u(z=0) -> interpolation (core/field.py Operand.__call__)."""
        from . import operators as ops
        out = self
        for name, pos in kw.items():
            coord = self.dist.get_coord(name)
            out = ops.Interpolate(out, coord, pos)
        return out

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        from . import operators as ops
        if method != "__call__" or kw:
            return NotImplemented
        if len(inputs) == 1:
            return ops.UnaryGridFunction(ufunc, inputs[0])
        if ufunc is np.multiply:
            return ops.Multiply(*inputs)
        if ufunc is np.add:
            return ops.Add(*inputs)
        if ufunc is np.subtract:
            return ops.Add(inputs[0], ops.Multiply(-1, inputs[1]))
        if ufunc is np.true_divide:
            a, b = inputs
            if isinstance(b, numbers.Number):
                return ops.Multiply(1.0 / b, a)
            return ops.Multiply(a, ops.Power(b, -1))
        if ufunc is np.power:
            return ops.Power(*inputs)
        if ufunc is np.matmul:
            return ops.DotProduct(*inputs)
        return NotImplemented

    @property
    def ncomp(self):
        n = 1
        for cs in self.tensorsig:
            n *= cs.dim
        return n

    @property
    def tshape(self):
        return tuple(cs.dim for cs in self.tensorsig)

    def evaluate(self):
        from .evaluator import evaluate_expression
        return evaluate_expression(self)


class Field(Operand):
    def __init__(self, dist, bases=None, name=None, tensorsig=None, dtype=None):
        if bases is None:
            bases = ()
        if not isinstance(bases, (tuple, list)):
            bases = (bases,)
        self.dist = dist
        self.name = name
        self.tensorsig = tuple(tensorsig) if tensorsig else ()
        self.dtype = np.dtype(dtype if dtype is not None else dist.dtype)
        if self.dtype != np.float64:
            raise NotImplementedError("only float64 fields are implemented in this round")
        self.domain = Domain(dist, bases)
        dist._register_domain(self.domain)
        if getattr(dist, "_pencil_geom", None) is None and dist._jacobi_axes | set(
                ax for ax, b in enumerate(self.domain.by_axis) if b is not None) == set(range(dist.dim)) and all(
                self.domain.by_axis[ax] is not None for ax in dist.separable_axes):
            from .evaluator import set_pencil_geom
            set_pencil_geom(dist, self.domain.by_axis)
        self.scales = (1.0,) * dist.dim
        # device arrays (storage order) and which one is current
        self._c = None
        self._g = None
        self._g_scales = None
        self.layout = "c"               # layout of the authoritative data
        # host mirror
        self._host = None
        self._host_layout = None
        self._host_scales = None
        self._authority = "device"      # 'device' | 'host'
        self._adopted = False           # coefficient array is a view into a solver's state vector
        self._tiled = None              # the SystemBuffer whose tile-major rows _c is a view of (SolverBase._enable_state_tiling)
        self._nrows = 0

    def __repr__(self):
        return self.name or "<Field %d>" % id(self)

    __str__ = __repr__

    # ---- shapes ----------------------------------------------------------------------------------
    def _user_shape(self, layout, scales):
        sp = self.domain.coeff_shape() if layout == "c" else self.domain.grid_shape(scales)
        return self.tshape + tuple(sp)

    def _storage_shape(self, layout, scales):
        sp = self.domain.storage_coeff_shape() if layout == "c" else self.domain.storage_grid_shape(scales)
        return (self.ncomp,) + tuple(sp)

    @property
    def global_shape(self):
        return self.domain.coeff_shape() if self.layout == "c" else self.domain.grid_shape(self.scales)

    def _remedy_scales(self, scales):
        if scales is None:
            return self.scales
        if isinstance(scales, numbers.Number):
            return (float(scales),) * self.dist.dim
        return tuple(float(s) for s in scales)

    # ---- host <-> storage order ---------------------------------------------------------------------
    def _to_storage(self, user_arr, layout, scales):
        nt = len(self.tensorsig)
        arr = np.asarray(user_arr, dtype=np.float64).reshape(self._user_shape(layout, scales))
        perm = tuple(range(nt)) + tuple(nt + ax for ax in self.dist.storage_order)
        return np.ascontiguousarray(np.transpose(arr, perm)).reshape(self._storage_shape(layout, scales))

    def _from_storage(self, st_arr, layout, scales):
        nt = len(self.tensorsig)
        sp = self.domain.storage_coeff_shape() if layout == "c" else self.domain.storage_grid_shape(scales)
        arr = np.asarray(st_arr).reshape(self.tshape + tuple(sp))
        inv = np.argsort(self.dist.storage_order)
        perm = tuple(range(nt)) + tuple(nt + int(i) for i in inv)
        return np.ascontiguousarray(np.transpose(arr, perm))

    # ---- device residency ---------------------------------------------------------------------------
    @property
    def ex(self):
        return self.dist.executor

    def _alloc_c(self):
        """The coefficient array to WRITE natural-layout data into (followed by _coeff_written)."""
        if self._tiled is not None:
            return self._tiled.natural_rows(self.ex, self._row0, self._nrows, current=False).reshape(
                self._storage_shape("c", None))
        if self._c is None:
            self._c = self.ex.zeros(self._storage_shape("c", None))
        return self._c

    def _coeff_written(self):
        """Natural coefficient data was written into _alloc_c(): a state field kept tile-major takes it over."""
        if self._tiled is not None:
            self._tiled.commit_rows(self.ex, self._row0, self._nrows)

    def _alloc_g(self, scales):
        shape = self._storage_shape("g", scales)
        if self._g is None or self._g_scales != scales:
            self._g = self.ex.empty(shape)
            self._g_scales = scales
        return self._g

    def _sync_to_device(self):
        """Upload the host mirror if the user may have touched it."""
        if self._authority != "host":
            return
        self._authority = "device"
        lay, sc = self._host_layout, self._host_scales
        st = self._to_storage(self._host, lay, sc)
        if lay == "c":
            self.ex.upload(self._alloc_c(), st)
            self._coeff_written()
        else:
            self.ex.upload(self._alloc_g(sc), st)
            self.scales = sc
        self.layout = lay

    def require_coeff_space(self):
        """Device coefficient array in the NATURAL layout, current (for a state field kept tile-major by its solver: the
        natural shadow of its rows, refreshed from the state when stale -- read-only for the caller)."""
        self._sync_to_device()
        if self.layout == "g":
            self.dist.transformer.forward(self, self._g, self._g_scales, self._alloc_c())
            self._coeff_written()
            self.layout = "c"
        elif self._c is None:
            self._alloc_c()
        if self._tiled is not None:
            return self._tiled.natural_rows(self.ex, self._row0, self._nrows).reshape(self._storage_shape("c", None))
        return self._c

    def coeff_tiled(self):
        """(device coefficient array as the solver stores it, row length of its tile-major rows or 0 for natural)."""
        if self._tiled is None:
            return self.require_coeff_space(), 0
        self._sync_to_device()
        if self.layout == "g":
            self.dist.transformer.forward(self, self._g, self._g_scales, self._alloc_c())
            self._coeff_written()
            self.layout = "c"
        return self._c, int(self._tiled.tiled)

    def require_grid_space(self, scales=None):
        self._sync_to_device()
        scales = self._remedy_scales(scales)
        if self.layout == "g" and self._g_scales == scales:
            self.scales = scales
            return self._g
        c = self.require_coeff_space()
        g = self._alloc_g(scales)
        self.dist.transformer.backward(self, c, g, scales)
        self.layout = "g"
        self.scales = scales
        return g

    def coeff_data(self):
        """Device coefficient array (storage order), current."""
        return self.require_coeff_space()

    def grid_data(self, scales=None):
        return self.require_grid_space(self.domain.dealias if scales is None else scales)

    def mark_device_coeff_current(self):
        """Called by the solver after it wrote the coefficient array on the device."""
        self.layout = "c"
        self._authority = "device"

    # ---- user access -----------------------------------------------------------------------------
    def change_scales(self, scales):
        scales = self._remedy_scales(scales)
        if scales == self.scales:
            return
        self._sync_to_device()
        if self.layout == "g":
            self.require_coeff_space()
        self.scales = scales

    preset_scales = change_scales

    def change_layout(self, layout):
        if layout == "c":
            self.require_coeff_space()
        else:
            self.require_grid_space(self.scales)

    def __getitem__(self, key):
        if isinstance(key, tuple):
            layout, scales = key
            self.change_scales(scales)
        else:
            layout = key
        layout = "c" if layout in ("c", "coeff") else "g"
        if not (self._authority == "host" and self._host_layout == layout
                and (layout == "c" or self._host_scales == self.scales)):
            self._sync_to_device()
            if layout == "c":
                dev = self.require_coeff_space()
            else:
                dev = self.require_grid_space(self.scales)
            self._host = self._from_storage(self.ex.download(dev), layout, self.scales)
            self._host_layout, self._host_scales = layout, self.scales
        # the caller may modify the mirror in place: the host copy is authoritative from now on
        self._authority = "host"
        return self._host

    def snapshot_async(self, layout):
        """Start copying the current data to the host without blocking the device (executor.download_async); returns a
        callable that yields the user-ordered array, or None when the executor has no asynchronous path.  Used by the
        file handlers: the copy overlaps the following timesteps."""
        ex = self.ex
        if not hasattr(ex, "download_async") or self._authority == "host":
            return None
        layout = "c" if layout in ("c", "coeff") else "g"
        scales = self.scales
        dev = self.require_coeff_space() if layout == "c" else self.require_grid_space(scales)
        pend = ex.download_async(dev)
        return lambda: self._from_storage(pend.wait(), layout, scales)

    def __setitem__(self, key, data):
        if isinstance(key, tuple):
            layout, scales = key
            self.scales = self._remedy_scales(scales)
        else:
            layout = key
        layout = "c" if layout in ("c", "coeff") else "g"
        shape = self._user_shape(layout, self.scales)
        if self._host is None or self._host.shape != shape or data is not self._host:
            host = np.empty(shape)
            host[...] = data
            self._host = host
        self._host_layout, self._host_scales = layout, self.scales
        self._authority = "host"

    @property
    def data(self):
        return self[self.layout if self._authority == "device" else self._host_layout]

    def copy(self):
        f = Field(self.dist, bases=self.domain.bases, tensorsig=self.tensorsig, name=self.name)
        f["c"] = self["c"].copy()
        return f

    def fill_random(self, layout=None, scales=None, seed=None, chunk_size=2 ** 20, distribution="standard_normal", **kw):
        """Reproducible random data: the same global stream the reference draws
        (core/field.py:898-943, tools/random_arrays.py:7-55: chunks of min(size, chunk_size) from
        default_rng(seed), C-ordered over (tensor components, global shape))."""
        if scales is not None:
            self.change_scales(scales)
        layout = layout or self.layout
        layout = "c" if layout in ("c", "coeff") else "g"
        gsp = self.domain.global_coeff_shape() if layout == "c" else self.domain.global_grid_shape(self.scales)
        shape = self.tshape + tuple(gsp)
        n = int(np.prod(shape))
        cs = min(n, chunk_size)
        rng = np.random.default_rng(seed)
        draw = getattr(rng, distribution)
        out = np.empty(n)
        pos = 0
        while pos < n:
            chunk = draw(size=cs, **kw)
            m = min(cs, n - pos)
            out[pos:pos + m] = chunk[:m]
            pos += m
        sl = (slice(None),) * len(self.tshape) + self.domain.local_slices(layout, self.scales)
        self[layout] = out.reshape(shape)[sl]

    def low_pass_filter(self, shape=None, scales=None):
        """Zero the modes above the given relative scales by a round trip through a coarser grid
        (core/field.py:945-967)."""
        original = self.scales
        if shape is not None:
            if scales is not None:
                raise ValueError("Specify either shape or scales.")
            full = self._user_shape("g", self._remedy_scales(1))[len(self.tensorsig):]
            scales = tuple(np.array(shape) / np.array(full))
        self.change_scales(scales)
        self.require_grid_space(self.scales)
        self.change_scales(original)

    def high_pass_filter(self, shape=None, scales=None):
        """Zero the modes below the given relative scales (core/field.py:969-993)."""
        data_orig = self["c"].copy()
        self.low_pass_filter(shape=shape, scales=scales)
        self["c"] = data_orig - self["c"]

    def allgather_data(self, layout=None):
        return self[layout or "g"].copy()

    # ---- linear-expression protocol -----------------------------------------------------------------
    def lin(self, ctx):
        from .polyop import LinExpr
        nz = self.dist.coupled_size(self.domain)
        return LinExpr.identity(self, self.ncomp, nz)

    def leaves(self):
        return {self}

    def has_dt(self):
        return False
