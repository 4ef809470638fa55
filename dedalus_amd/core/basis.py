"""
Spectral bases: sizes, grids, wavenumbers and the per-axis operator matrices.

Host-side metadata only; all arithmetic on field data runs through libdedalus_hip.so.  The public
names and constructor arguments follow dedalus/core/basis.py (RealFourier :1104-1135,
Jacobi :432-640, ChebyshevT/U/V, Legendre, Ultraspherical :616-640).
"""

import numpy as np
from scipy import sparse

from ..tools import jacobi


def _grid_args(args, scale):
    """(dist or None, scale) from global_grid(scale) / global_grid(dist, scale) / global_grid(dist, scale=...)"""
    dist = None
    args = list(args)
    if args and not isinstance(args[0], (int, float)):
        dist = args.pop(0)
    if args:
        scale = args[0]
    return dist, scale


class Basis:
    dim = 1

    def grid_size(self, scale):
        if self.size == 1:
            return 1
        return int(np.ceil(scale * self.size))

    def __radd__(self, other):
        return self.__add__(other)

    def __rmul__(self, other):
        return self.__mul__(other)

    def __eq__(self, other):
        return type(self) is type(other) and self._key == other._key

    def __hash__(self):
        return hash((type(self).__name__, self._key))

    def __ne__(self, other):
        return not self.__eq__(other)


class RealFourier(Basis):
    """Real sine/cosine basis, modes [cos 0x, -sin 0x, cos 1x, -sin 1x, ...] (basis.py:1104-1110)."""

    separable = True
    kind = "rf"

    def __init__(self, coord, size, bounds, dealias=1, library=None):
        size = int(size)
        if size % 2:
            raise ValueError("RealFourier size must be even.")
        self.coord, self.size, self.bounds = coord, size, tuple(float(b) for b in bounds)
        self.dealias = float(dealias[0]) if isinstance(dealias, (tuple, list)) else float(dealias)
        self.library = library
        self.length = self.bounds[1] - self.bounds[0]
        self.coeff_size = size
        self.constant_mode_value = 1.0
        self._key = (coord, size, self.bounds, self.dealias)

    def __repr__(self):
        return "RealFourier(%s, %d)" % (self.coord.name, self.size)

    @property
    def mode_wavenumbers(self):
        """Physical wavenumber of each (cos, msin) pair: k_m = 2 pi m / L, m < size/2 (Nyquist excluded)."""
        return 2 * np.pi * np.arange(self.size // 2) / self.length

    @property
    def wavenumbers(self):
        return np.repeat(self.mode_wavenumbers, 2)

    def global_grid(self, *args, scale=1):
        """global_grid(scale) or, with the reference's signature, global_grid(dist, scale) (reshaped to the
        distributor's dimensions, core/basis.py:362-366)."""
        dist, scale = _grid_args(args, scale)
        N = self.grid_size(scale)
        g = self.bounds[0] + self.length * np.arange(N) / N
        return g if dist is None else dist._reshape_axis(g, dist.coord_axis(self.coord))

    def grid_spacing(self, scale=1):
        N = self.grid_size(scale)
        return np.full(N, self.length / N)

    def plan_spec(self, scale):
        return ("rfft", self.grid_size(scale), self.size)

    def derivative_basis(self, order=1):
        return self

    def __add__(self, other):
        if other is None or other == self:
            return self
        return NotImplemented

    def __mul__(self, other):
        if other is None or other == self:
            return self
        return NotImplemented


class Jacobi(Basis):
    """Jacobi polynomial basis on the (a0, b0) Gauss grid (basis.py:432-640)."""

    separable = False
    kind = "jac"

    def __init__(self, coord, size, bounds, a, b, a0=None, b0=None, dealias=1, library=None):
        self.coord, self.size, self.bounds = coord, int(size), tuple(float(x) for x in bounds)
        self.a, self.b = float(a), float(b)
        self.a0 = self.a if a0 is None else float(a0)
        self.b0 = self.b if b0 is None else float(b0)
        self.dealias = float(dealias[0]) if isinstance(dealias, (tuple, list)) else float(dealias)
        self.library = library
        self.length = self.bounds[1] - self.bounds[0]
        self.stretch = self.length / 2.0           # dz_problem / dz_native
        self.coeff_size = self.size
        self.constant_mode_value = 1.0 / np.sqrt(jacobi.mass(self.a, self.b))
        self._key = (coord, self.size, self.bounds, self.a, self.b, self.a0, self.b0, self.dealias)

    def __repr__(self):
        return "Jacobi(%s, %d, a=%g, b=%g)" % (self.coord.name, self.size, self.a, self.b)

    def clone_with(self, **kw):
        args = dict(coord=self.coord, size=self.size, bounds=self.bounds, a=self.a, b=self.b, a0=self.a0,
                    b0=self.b0, dealias=self.dealias, library=self.library)
        args.update(kw)
        return Jacobi(**args)

    def derivative_basis(self, order=1):
        return self.clone_with(a=self.a + order, b=self.b + order)

    def _same_grid(self, other):
        return (isinstance(other, Jacobi) and self.coord == other.coord and self.bounds == other.bounds
                and self.a0 == other.a0 and self.b0 == other.b0 and self.dealias == other.dealias)

    def __add__(self, other):
        if other is None or other == self:
            return self
        if self._same_grid(other):
            return self.clone_with(size=max(self.size, other.size), a=max(self.a, other.a), b=max(self.b, other.b))
        return NotImplemented

    def __mul__(self, other):
        # products are formed on the grid and land in the grid basis (basis.py:536-549)
        if other is None or other == self:
            return self
        if self._same_grid(other):
            return self.clone_with(size=max(self.size, other.size), a=self.a0, b=self.b0)
        return NotImplemented

    # ---- grids -------------------------------------------------------------------------------
    def native_grid(self, scale=1):
        return jacobi.build_grid(self.grid_size(scale), self.a0, self.b0)

    def global_grid(self, *args, scale=1):
        """global_grid(scale) or, with the reference's signature, global_grid(dist, scale) (reshaped to the
        distributor's dimensions, core/basis.py:362-366)."""
        dist, scale = _grid_args(args, scale)
        g = self.bounds[0] + (self.native_grid(scale) + 1.0) * self.stretch
        return g if dist is None else dist._reshape_axis(g, dist.coord_axis(self.coord))

    def grid_spacing(self, scale=1):
        return np.gradient(self.global_grid(scale), edge_order=2) if self.grid_size(scale) > 2 else \
            np.full(self.grid_size(scale), self.length)

    # ---- transform description -----------------------------------------------------------------
    def conversion_bands(self):
        """(offsets, bands[d][k] = C[k, k+offset_d]) of the grid-basis -> this-basis conversion."""
        if self.a == self.a0 and self.b == self.b0:
            return (), None
        cached = getattr(self, "_conv_bands", None)
        if cached is not None:
            return cached
        conv = jacobi.conversion_matrix(self.size, self.a0, self.b0, self.a, self.b).toarray()
        offs = [o for o in range(self.size) if np.any(np.diagonal(conv, o) != 0)]
        bands = np.zeros((len(offs), self.size))
        for d, o in enumerate(offs):
            bands[d, :self.size - o] = np.diagonal(conv, o)
        self._conv_bands = (tuple(offs), bands)       # (called for every transform of every step)
        return self._conv_bands

    def plan_spec(self, scale):
        cache = self.__dict__.setdefault("_plan_specs", {})
        if scale in cache:
            return cache[scale]
        N = self.grid_size(scale)
        if self.a0 == self.b0 == -0.5 and self.library in (None, "fftw_dct", "scipy_dct", "hip"):
            offs, bands = self.conversion_bands()
            key = bands.tobytes() if bands is not None else b""
            spec = ("cheb", N, self.size, offs, key)
        else:
            spec = ("mmt", N, self.size, self.a, self.b, self.a0, self.b0)
        cache[scale] = spec
        return spec

    def mmt_matrices(self, N):
        """JacobiMMT definition (transforms.py:118-158): forward (size x N), backward (N x size)."""
        M = self.size
        z, w = jacobi.quadrature(N, self.a0, self.b0)
        Mk = min(N, M)
        P0 = jacobi.polynomials(Mk, self.a0, self.b0, z)
        fwd = np.zeros((M, N))
        fwd[:Mk] = P0 * w
        conv = jacobi.conversion_matrix(M, self.a0, self.b0, self.a, self.b)
        fwd = conv @ fwd
        bwd = np.zeros((N, M))
        bwd[:, :Mk] = jacobi.polynomials(Mk, self.a, self.b, z).T
        return np.ascontiguousarray(fwd), np.ascontiguousarray(bwd)

    # ---- operator matrices along this axis (all act on coefficient vectors) ---------------------
    def convert_matrix(self, out_basis):
        """ConvertJacobi (basis.py:643-657)."""
        return jacobi.conversion_matrix(self.size, self.a, self.b, out_basis.a, out_basis.b)

    def differentiate_matrix(self):
        """DifferentiateJacobi (basis.py:679-697)."""
        return jacobi.differentiation_matrix(self.size, self.a, self.b) / self.stretch

    def interpolate_vector(self, position):
        """InterpolateJacobi (basis.py:700-722)."""
        zn = (position - self.bounds[0]) / self.stretch - 1.0
        return jacobi.interpolation_vector(self.size, self.a, self.b, zn)

    def integrate_vector(self):
        """IntegrateJacobi (basis.py:725-744)."""
        return jacobi.integration_vector(self.size, self.a, self.b) * self.stretch


def Legendre(*args, **kw):
    return Jacobi(*args, a=0, b=0, **kw)


def Ultraspherical(*args, alpha, alpha0=None, **kw):
    if alpha0 is None:
        alpha0 = alpha
    return Jacobi(*args, a=alpha - 0.5, b=alpha - 0.5, a0=alpha0 - 0.5, b0=alpha0 - 0.5, **kw)


def ChebyshevT(*args, **kw):
    return Ultraspherical(*args, alpha=0, **kw)


def ChebyshevU(*args, **kw):
    return Ultraspherical(*args, alpha=1, **kw)


def ChebyshevV(*args, **kw):
    return Ultraspherical(*args, alpha=2, **kw)


Chebyshev = ChebyshevT
