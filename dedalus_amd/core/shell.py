"""
Spherical shells: SphericalCoordinates, ShellBasis and tensor fields with their device transforms
(the field-level transform path of BASELINE config 5).

Reference sequence (core/basis.py): backward = ShellBasis.backward_transform_radius (:4488-4508: radial Jacobi
transform per regularity component, regularity -> spin recombination Q(ell), radial factor (dR/r)^k), then the
colatitude step of the angular basis (SWSH per spin weight + spin -> coordinate recombination, :3134-3153, with
the 3 x 3 U of SphericalCoordinates core/coords.py:338-351), then the azimuthal FFT; forward is the mirror.

On the device a field is [component][2 m + part][ell][n] with n the radial index (contiguous); the angular
(m, ell) bookkeeping and the user-facing packed coefficient layout are those of core/sphere.py.  Every step is one
launch over the whole field: ddh_cheb_* / ddh_mmt_apply (radius), ddh_regularity_recombine with the radial
factor fused, ddh_grouped_mmt_* with the radial points as GEMM columns, ddh_spin_recombine, ddh_rfft_*.
Operators and the per-ell radial solves of the shell are not built yet.
"""

import numbers

import numpy as np

from ..tools import sphere as sph
from . import curvilinear
from .basis import Jacobi
from .coords import Coordinate
from .sphere import S2Coordinates, SphereBasis


class SphericalCoordinates:
    """(azimuth, colatitude, radius); spin and regularity component ordering (-, +, 0)  (core/coords.py:300-390)."""
    dim = 3
    spin_ordering = (-1, +1, 0)
    reg_ordering = (-1, +1, 0)

    def __init__(self, azimuth, colatitude, radius):
        self.names = (azimuth, colatitude, radius)
        self.azimuth = Coordinate(azimuth, cs=self)
        self.colatitude = Coordinate(colatitude, cs=self)
        self.radius = Coordinate(radius, cs=self)
        self.coords = (self.azimuth, self.colatitude, self.radius)
        self.S2coordsys = S2Coordinates(azimuth, colatitude)

    def __iter__(self):
        return iter(self.coords)

    @staticmethod
    def U_forward(order=1):
        """coordinate (phi, theta, r) -> spin (-, +, 0) components: u[+-] = (u[theta] +- i u[phi]) / sqrt 2, u[0] = u[r]"""
        U = np.array([[-1j, 1, 0], [+1j, 1, 0], [0, 0, np.sqrt(2)]]) / np.sqrt(2)
        out = np.array([[1.0 + 0j]])
        for _ in range(order):
            out = np.kron(out, U)
        return out


class ShellBasis:
    """Shell: SWSH in angle x Jacobi(alpha + k) in radius on [Ri, Ro]  (core/basis.py:4380-4440, 3682-3816)."""

    def __init__(self, coordsys, shape, dtype=np.float64, radii=(1, 2), alpha=(-0.5, -0.5), dealias=(1, 1, 1), k=0,
                 azimuth_library=None, colatitude_library=None, radius_library=None):
        if not isinstance(coordsys, SphericalCoordinates):
            raise ValueError("Shell coordsys must be SphericalCoordinates.")
        if np.dtype(dtype) != np.float64:
            raise NotImplementedError("shell fields: float64 only")
        if min(radii) <= 0:
            raise ValueError("Radii must be positive.")
        if isinstance(dealias, numbers.Number):
            dealias = (dealias,) * 3
        self.coordsys, self.shape, self.dtype = coordsys, tuple(int(s) for s in shape), np.dtype(dtype)
        self.radii, self.alpha, self.k = tuple(radii), tuple(alpha), int(k)
        self.dealias = tuple(float(d) for d in dealias)
        self.Nr = self.shape[2]
        self.dR = self.radii[1] - self.radii[0]
        self.rho = (self.radii[1] + self.radii[0]) / self.dR
        self.sphere = SphereBasis(coordsys.S2coordsys, self.shape[:2], dtype=dtype, radius=1, dealias=self.dealias[:2])
        # radial Jacobi basis: coefficients in (alpha + k), grid of alpha
        self.radial = Jacobi(coordsys.radius, size=self.Nr, bounds=self.radii, a=alpha[0] + k, b=alpha[1] + k,
                             a0=alpha[0], b0=alpha[1], dealias=self.dealias[2])
        self._plans = {}

    @property
    def outer_surface(self):
        return SphereBasis(self.coordsys.S2coordsys, self.shape[:2], radius=self.radii[1], dealias=self.dealias[:2])

    @property
    def inner_surface(self):
        return SphereBasis(self.coordsys.S2coordsys, self.shape[:2], radius=self.radii[0], dealias=self.dealias[:2])

    def grid_shape(self, scales):
        return self.sphere.grid_shape(scales[:2]) + (int(np.ceil(scales[2] * self.Nr)),)

    def grids(self, scales):
        phi, theta = self.sphere.grids(scales[:2])
        return phi, theta, self.radius_grid(scales[2])

    def radius_grid(self, scale):
        from ..tools import jacobi
        N = int(np.ceil(scale * self.Nr))
        z, _ = jacobi.quadrature(N, self.alpha[0], self.alpha[1])
        return (self.dR / 2 * (np.asarray(z, dtype=np.float64) + self.rho))

    @staticmethod
    def spin_indices(rank):
        return list(np.ndindex(*((3,) * rank)))

    @staticmethod
    def spin_totals(rank):
        return [sum((-1, +1, 0)[a] for a in idx) for idx in np.ndindex(*((3,) * rank))]

    def recombination_matrix(self, rank, forward):
        U = SphericalCoordinates.U_forward(rank)
        if not forward:
            U = U.T.conj()
        return np.kron(U.real, np.eye(2)) + np.kron(U.imag, np.array([[0.0, -1.0], [1.0, 0.0]]))

    def colatitude_plan(self, ex, Ntheta_g, rank):
        """grouped SWSH plan over all (component, m) of a rank-`rank` tensor (as SphereBasis.colatitude_plan)"""
        key = ("swsh", id(ex), Ntheta_g, rank)
        if key not in self._plans:
            sb = self.sphere
            groups, keys, fwd, bwd, cache = [], [], [], [], {}
            for i, s in enumerate(self.spin_totals(rank)):
                for m in range(sb.nm):
                    mk = m + 4096 * (s + 8)
                    ne = max(sb.Lmax + 1 - m, 0)
                    groups.append((mk if ne > 0 else -1 - m, i * 2 * sb.nm + 2 * m, i * 2 * sb.nm + 2 * m, 2, m, 1, ne))
                    if ne > 0 and mk not in cache:
                        cache[mk] = sph.swsh_matrices(Ntheta_g, sb.Lmax, m, s)
                        keys.append(mk)
                        fwd.append(cache[mk][0])
                        bwd.append(cache[mk][1])
            self._plans[key] = ex.make_grouped_mmt(Ntheta_g, np.array(groups, dtype=np.int64), keys, fwd, bwd)
        return self._plans[key]

    def regularity_plan(self, ex, rank):
        """Q(ell) tables on the natural (2 m + part, ell) slots.  Built from the reference's ell_maps of the packed
        layout (bounding boxes that overlap: a slot covered by several boxes is recombined by each of them in
        turn, core/basis.py:3595-3626) and gathered to the natural slots, so every mode sees exactly the
        matrix product the reference applies to it."""
        key = ("reg", id(ex), rank)
        if key not in self._plans:
            sb = self.sphere
            slot_p, fwd, bwd = curvilinear.recombination_tables(sb.packed_ell_rows(), sb.packed_shape(), rank)
            rows, cols, ok = sb.pack_index()
            slot = -np.ones((2 * sb.nm, sb.nl), dtype=np.int32)
            slot[rows[ok], cols[ok]] = slot_p[ok]
            self._plans[key] = (ex.make_recombination(slot, fwd) if rank > 0 else None,
                                ex.make_recombination(slot, bwd) if rank > 0 else None)
        return self._plans[key]

    def radial_factor(self, ex, scale, power):
        key = ("fac", id(ex), scale, power)
        if key not in self._plans:
            r = self.radius_grid(scale)
            self._plans[key] = ex.from_host(np.ascontiguousarray((self.dR / r) ** power))
        return self._plans[key]


class ShellDistributor:
    def __init__(self, coordsys, comm=None, mesh=None, dtype=None, executor=None):
        self.coordsystems = (coordsys,)
        self.coordsys = coordsys
        self.coords = coordsys.coords
        self.dim = 3
        self.dtype = np.dtype(np.float64 if dtype is None else dtype)
        if mesh is not None and int(np.prod(mesh)) > 1:
            raise NotImplementedError("shell fields live on one device in this round")
        self.mesh, self.size, self.rank, self.comm = (), 1, 0, comm
        self._executor = executor

    @property
    def executor(self):
        if self._executor is None:
            from ..executor import HipExecutor
            self._executor = HipExecutor()       # raises without a gfx950 device: no CPU fallback
        return self._executor

    def Field(self, name=None, bases=None, tensorsig=None, dtype=None):
        if isinstance(bases, (tuple, list)):
            bases = bases[0] if bases else None
        if not isinstance(bases, ShellBasis):
            raise NotImplementedError("fields in spherical coordinates need a ShellBasis in this round")
        return ShellField(self, bases, rank=len(tensorsig) if tensorsig else 0, name=name)

    ScalarField = Field

    def VectorField(self, coordsys, name=None, bases=None, dtype=None):
        return self.Field(name=name, bases=bases, tensorsig=(coordsys,))

    def TensorField(self, coordsys, name=None, bases=None, order=2, dtype=None):
        sig = tuple(coordsys) if isinstance(coordsys, (tuple, list)) else (coordsys,) * order
        return self.Field(name=name, bases=bases, tensorsig=sig)

    def local_grids(self, *bases, scales=None):
        basis = bases[0]
        if scales is None:
            scales = (1, 1, 1)
        elif isinstance(scales, numbers.Number):
            scales = (scales,) * 3
        phi, theta, r = basis.grids(scales)
        return phi[:, None, None], theta[None, :, None], r[None, None, :]


def backward(dist, basis, rank, c, scales):
    """coefficients [nc][2 nm][nl][Nr] (regularity components) -> grid [nc][Nphi_g][Ntheta_g][Nr_g] (coordinate components)"""
    ex = dist.executor
    sb = basis.sphere
    nc = 3 ** rank
    Np, Nt, Ng = basis.grid_shape(scales)
    nslots = nc * 2 * sb.nm * sb.nl
    t0 = ex.empty((nc, 2 * sb.nm, sb.nl, Ng))
    ex.transform(basis.radial.plan_spec(scales[2]), basis.radial, "backward", c, t0, nslots, 1)
    ex.regularity_recombine(t0, basis.regularity_plan(ex, rank)[1],
                            basis.radial_factor(ex, scales[2], basis.k) if basis.k > 0 else None)
    t1 = ex.empty((nc, 2 * sb.nm, Nt, Ng))
    basis.colatitude_plan(ex, Nt, rank).backward(t0.reshape(1, nc * 2 * sb.nm, sb.nl, Ng),
                                                 t1.reshape(1, nc * 2 * sb.nm, Nt, Ng))
    if rank > 0:
        t2 = ex.empty((nc, 2 * sb.nm, Nt * Ng))
        ex.spin_recombine(t1.reshape(nc, 2 * sb.nm, Nt * Ng), t2, basis.recombination_matrix(rank, forward=False))
    else:
        t2 = t1
    g = ex.empty((nc, Np, Nt, Ng))
    ex.transform(("rfft", Np, sb.Nphi), None, "backward", t2, g, nc, Nt * Ng)
    return g


def forward(dist, basis, rank, g, scales):
    ex = dist.executor
    sb = basis.sphere
    nc = 3 ** rank
    Np, Nt, Ng = basis.grid_shape(scales)
    t1 = ex.empty((nc, 2 * sb.nm, Nt * Ng))
    ex.transform(("rfft", Np, sb.Nphi), None, "forward", g, t1, nc, Nt * Ng)
    if rank > 0:
        t2 = ex.empty((nc, 2 * sb.nm, Nt * Ng))
        ex.spin_recombine(t1, t2, basis.recombination_matrix(rank, forward=True))
    else:
        t2 = t1
    t3 = ex.zeros((nc, 2 * sb.nm, sb.nl, Ng))
    basis.colatitude_plan(ex, Nt, rank).forward(t2.reshape(1, nc * 2 * sb.nm, Nt, Ng),
                                                t3.reshape(1, nc * 2 * sb.nm, sb.nl, Ng))
    ex.regularity_recombine(t3, basis.regularity_plan(ex, rank)[0],
                            basis.radial_factor(ex, scales[2], -basis.k) if basis.k > 0 else None)
    c = ex.empty((nc, 2 * sb.nm, sb.nl, basis.Nr))
    ex.transform(basis.radial.plan_spec(scales[2]), basis.radial, "forward", t3, c, nc * 2 * sb.nm * sb.nl, 1)
    return c


class ShellField:
    """Tensor field on a ShellBasis: device-resident data with lazily synchronised host mirrors, `['g']` / `['c']`
    in the reference's shapes (packed (m, ell) coefficient layout, core/basis.py:2839-2891)."""

    def __init__(self, dist, basis, rank=0, name=None):
        self.dist, self.basis, self.rank, self.name = dist, basis, rank, name
        self.scales = (1.0, 1.0, 1.0)
        self._c = None
        self._g = None
        self._g_scales = None
        self.layout = "c"
        self._host = None
        self._host_layout = None
        self._host_scales = None
        self._authority = "device"

    @property
    def ncomp(self):
        return 3 ** self.rank

    @property
    def tensorsig(self):
        return (self.dist.coordsys,) * self.rank

    @property
    def ex(self):
        return self.dist.executor

    def _cshape(self):
        sb = self.basis.sphere
        return (self.ncomp, 2 * sb.nm, sb.nl, self.basis.Nr)

    def _user_shape(self, layout, scales):
        t = (3,) * self.rank
        if layout == "g":
            return t + self.basis.grid_shape(scales)
        return t + self.basis.sphere.packed_shape() + (self.basis.Nr,)

    def _remedy(self, scales):
        if scales is None:
            return (1.0,) * 3
        if isinstance(scales, numbers.Number):
            return (float(scales),) * 3
        return tuple(float(s) for s in scales)

    def _sync_to_device(self):
        if self._authority != "host":
            return
        self._authority = "device"
        lay, sc = self._host_layout, self._host_scales
        if lay == "c":
            rows, cols, ok = self.basis.sphere.pack_index()
            nat = np.zeros(self._cshape())
            h = self._host.reshape((self.ncomp,) + rows.shape + (self.basis.Nr,))
            for c in range(self.ncomp):
                nat[c][rows[ok], cols[ok], :] = h[c][ok]
            if self._c is None:
                self._c = self.ex.zeros(self._cshape())
            self.ex.upload(self._c, nat)
            self.layout = "c"
        else:
            shape = (self.ncomp,) + self.basis.grid_shape(sc)
            if self._g is None or self._g_scales != sc:
                self._g = self.ex.empty(shape)
                self._g_scales = sc
            self.ex.upload(self._g, np.ascontiguousarray(self._host.reshape(shape)))
            self.layout = "g"
            self.scales = sc

    def require_coeff_space(self):
        self._sync_to_device()
        if self.layout == "g":
            self._c = forward(self.dist, self.basis, self.rank, self._g, self._g_scales)
            self.layout = "c"
        if self._c is None:
            self._c = self.ex.zeros(self._cshape())
        return self._c

    def require_grid_space(self, scales=None):
        self._sync_to_device()
        scales = self._remedy(scales)
        if self.layout == "g" and self._g_scales == scales:
            return self._g
        c = self.require_coeff_space()
        self._g = backward(self.dist, self.basis, self.rank, c, scales)
        self._g_scales = scales
        self.layout = "g"
        self.scales = scales
        return self._g

    def change_scales(self, scales):
        scales = self._remedy(scales)
        if scales == self.scales:
            return
        self._sync_to_device()
        if self.layout == "g":
            self.require_coeff_space()
        self.scales = scales

    preset_scales = change_scales

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
            shape = self._user_shape(layout, self.scales)
            if layout == "c":
                nat = np.asarray(self.ex.download(self.require_coeff_space()))
                rows, cols, ok = self.basis.sphere.pack_index()
                out = np.zeros((self.ncomp,) + rows.shape + (self.basis.Nr,))
                for c in range(self.ncomp):
                    out[c][ok] = nat[c][rows[ok], cols[ok], :]
                self._host = out.reshape(shape)
            else:
                self._host = np.array(self.ex.download(self.require_grid_space(self.scales))).reshape(shape)
            self._host_layout, self._host_scales = layout, self.scales
        self._authority = "host"
        return self._host

    def __setitem__(self, key, data):
        if isinstance(key, tuple):
            layout, scales = key
            self.scales = self._remedy(scales)
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
