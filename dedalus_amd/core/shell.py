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

import logging
import numbers
import time as _time

import numpy as np

from ..tools import sphere as sph
from . import curvilinear
from .basis import Jacobi
from .coords import Coordinate
from .ivp_common import IVPLifecycle
from .sphere import S2Coordinates, SphereBasis


logger = logging.getLogger(__name__)


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
        return self.S2_basis(self.radii[1])

    @property
    def inner_surface(self):
        return self.S2_basis(self.radii[0])

    @property
    def radial_basis(self):
        """Tag for fields that depend on the radius only (NCCs: er, rvec; core/basis.py ShellRadialBasis)."""
        key = ("radial",)
        if key not in self._plans:
            self._plans[key] = RadialBasis(self)
        return self._plans[key]

    def S2_basis(self, radius=1):
        key = ("surf", float(radius))
        if key not in self._plans:
            self._plans[key] = SurfaceBasis(self, radius)
        return self._plans[key]

    def derivative_basis(self, order=1):
        return self.clone_with(k=self.k + order)

    def clone_with(self, **kw):
        args = dict(shape=self.shape, radii=self.radii, alpha=self.alpha, dealias=self.dealias, k=self.k)
        args.update(kw)
        key = ("clone", args["k"], tuple(args["shape"]))
        root = getattr(self, "_root", self)
        if key not in root._plans:
            b = ShellBasis(self.coordsys, dtype=self.dtype, **args)
            b._root = root
            if tuple(args["shape"]) == tuple(root.shape):
                b.sphere = root.sphere
            root._plans[key] = b
        return root._plans[key]

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
        store = getattr(self, "_root", self)._plans        # shared by the derivative bases (same angular part)
        if key not in store:
            sb = self.sphere
            groups, keys, fwd, bwd, cache = [], [], [], [], {}
            for i, s in enumerate(self.spin_totals(rank)):
                for ml in range(sb.nml):
                    m = sb.m0 + ml
                    mk = m + 4096 * (s + 8)
                    ne = max(sb.Lmax + 1 - m, 0)
                    row = i * 2 * sb.nml + 2 * ml
                    groups.append((mk if ne > 0 else -1 - m, row, row, 2, m, 1, ne))
                    if ne > 0 and mk not in cache:
                        cache[mk] = sph.swsh_matrices(Ntheta_g, sb.Lmax, m, s)
                        keys.append(mk)
                        fwd.append(cache[mk][0])
                        bwd.append(cache[mk][1])
            store[key] = ex.make_grouped_mmt(Ntheta_g, np.array(groups, dtype=np.int64), keys, fwd, bwd)
        return store[key]

    def regularity_plan(self, ex, rank):
        """Q(ell) tables on the natural (2 m + part, ell) slots.  Built from the reference's ell_maps of the packed
        layout (bounding boxes that overlap: a slot covered by several boxes is recombined by each of them in
        turn, core/basis.py:3595-3626) and gathered to the natural slots, so every mode sees exactly the
        matrix product the reference applies to it."""
        key = ("reg", id(ex), rank)
        store = getattr(self, "_root", self)._plans
        if key not in store:
            sb = self.sphere
            slot_p, fwd, bwd = curvilinear.recombination_tables(sb.packed_ell_rows(), sb.packed_shape(), rank)
            rows, cols, ok = sb.pack_index()
            slot = -np.ones((2 * sb.nm, sb.nl), dtype=np.int32)
            slot[rows[ok], cols[ok]] = slot_p[ok]
            slot = np.ascontiguousarray(slot[2 * sb.m0:2 * (sb.m0 + sb.nml)])
            store[key] = (ex.make_recombination(slot, fwd) if rank > 0 else None,
                          ex.make_recombination(slot, bwd) if rank > 0 else None)
        return store[key]

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
        P = int(np.prod(mesh)) if mesh is not None and len(tuple(mesh)) else 1
        self.mesh, self.size, self.rank, self.comm, self.pcomm = (), 1, 0, comm, None
        if P > 1:
            # one process per GPU on a 1-D mesh: azimuthal wavenumbers block-distributed in coefficient space,
            # colatitudes in grid space (the reference's layouts for a 1-D mesh, core/distributor.py:60-70)
            from ..parallel import Comm
            self.pcomm = Comm(P)
            self.mesh, self.size, self.rank = (P,), P, self.pcomm.rank
        coordsys.dist = self
        coordsys.S2coordsys.dist = self
        self._executor = executor

    def theta_range(self, Nt):
        """(first local colatitude index, number of local colatitudes) of a grid with Nt colatitudes"""
        if Nt % self.size:
            raise ValueError("%d colatitudes do not divide over %d ranks" % (Nt, self.size))
        n = Nt // self.size
        return self.rank * n, n

    @property
    def executor(self):
        if self._executor is None:
            from ..executor import HipExecutor
            self._executor = HipExecutor()       # raises without a gfx950 device: no CPU fallback
        return self._executor

    def Field(self, name=None, bases=None, tensorsig=None, dtype=None):
        if isinstance(bases, (tuple, list)):
            bases = bases[0] if bases else None
        rank = len(tensorsig) if tensorsig else 0
        if bases is None:
            if rank:
                raise NotImplementedError("constant tensor fields in spherical coordinates")
            return ConstField(self, name=name)
        if isinstance(bases, RadialBasis):
            return RadialField(self, bases.shell, rank=rank, name=name)
        if not isinstance(bases, (ShellBasis, SurfaceBasis)):
            raise NotImplementedError("fields in spherical coordinates need a ShellBasis, one of its surfaces or its radial basis")
        return ShellField(self, bases, rank=rank, name=name)

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
        t0, nt = self.theta_range(theta.size)
        return phi[:, None, None], theta[None, t0:t0 + nt, None], r[None, None, :]


class RadialBasis:
    def __init__(self, shell):
        self.shell = shell
        self.k = shell.k


class SurfaceBasis:
    """A SphereBasis used by fields of a SphericalCoordinates distributor (shell.outer_surface / inner_surface, tau
    fields): 3^rank spin components (-, +, 0) per (m, ell), one radial point (core/basis.py SphereBasis with a
    SphericalCoordinates coordsys, :2683-2686)."""

    def __init__(self, shell, radius):
        self.shell, self.radius = shell, float(radius)
        self.sphere = shell.sphere
        self.dealias = shell.dealias
        self.Nr = 1
        self.k = 0
        self.coordsys = shell.coordsys

    def grid_shape(self, scales):
        return self.sphere.grid_shape(scales[:2]) + (1,)

    def grids(self, scales):
        phi, theta = self.sphere.grids(scales[:2])
        return phi, theta, np.array([self.radius])

    def colatitude_plan(self, ex, Nt, rank):
        return self.shell.colatitude_plan(ex, Nt, rank)

    def recombination_matrix(self, rank, forward):
        return self.shell.recombination_matrix(rank, forward)

    def __eq__(self, other):
        return isinstance(other, SurfaceBasis) and other.shell is self.shell and other.radius == self.radius

    def __hash__(self):
        return hash((id(self.shell), self.radius))


def _radial_gemm(dist, basis, scale, nc, forward_dir):
    """The radial Jacobi transform of all (component, m, ell) lines as ONE dense GEMM with the (Ng x Ng, zero-padded)
    transform matrix (JacobiMMT, core/transforms.py:114-158) shared by every ell: `ddh_ell_terms_apply` runs it on the
    FP64 MFMA path.  OPT-IN (DDH_RADIAL_GEMM=1 when the radial grid size tiles by 64, =2 always): measured at config
    H it is slower than the LDS FFT kernel (step 13.4 ms against 10.3 ms) -- the per-ell GEMM tiles are mostly empty at
    low ell and the padded copies cost two more passes -- so the FFT path stays the default; the parity tests keep the
    matrix path alive as the independent check of the fast transform (the reference's own fast-vs-matrix test)."""
    import os
    mode = int(os.environ.get("DDH_RADIAL_GEMM", "0"))
    Ng = basis.grid_shape((1.0, 1.0, scale))[2]
    if mode == 0 or dist.size > 1 or Ng < basis.Nr or (mode == 1 and (Ng % 64 != 0 or basis.Nr < 64)):
        return None
    ex = dist.executor
    key = ("rgemm", id(ex), scale, nc, forward_dir)
    if key not in basis._plans:
        sb = basis.sphere
        fwd, bwd = basis.radial.mmt_matrices(Ng)              # (Nr x Ng), (Ng x Nr)
        A = np.zeros((1, Ng, Ng))
        if forward_dir:
            A[0, :basis.Nr, :] = fwd
        else:
            A[0, :, :basis.Nr] = bwd
        live = (np.arange(2 * sb.nml)[:, None] // 2) <= np.arange(sb.nl)[None, :]
        slot_map = np.where(live, 0, -1).astype(np.int32)
        basis._plans[key] = ex.make_ell_terms(sb.nml, sb.nl, Ng, nc, [(c, c, A) for c in range(nc)], slot_map)
    return basis._plans[key]


def backward(dist, basis, rank, c, scales):
    """coefficients [nc][2 nml][nl][Nr] (regularity components, local m) -> grid [nc][Nphi_g][Ntheta_g / P][Nr_g]
    (coordinate components, local colatitudes).  On P ranks the only exchange is the all-to-all between "m local
    block, all colatitudes" and "all m, colatitude block" that replaces the reference's (azimuth, colatitude)
    transpose (core/distributor.py:770-924), placed before the azimuthal FFT."""
    ex = dist.executor
    sb = basis.sphere
    nc = 3 ** rank
    nml = sb.nml
    Np, Nt, Ng = basis.grid_shape(scales)
    nslots = nc * 2 * nml * sb.nl
    if isinstance(basis, SurfaceBasis):
        t0 = c                                             # spin components, no radial axis
    else:
        t0 = ex.empty((nc, 2 * nml, sb.nl, Ng))
        gemm = _radial_gemm(dist, basis, scales[2], nc, False)
        if gemm is not None:
            gemm.apply(_padded(ex, c, Ng), t0)
        else:
            ex.transform(basis.radial.plan_spec(scales[2]), basis.radial, "backward", c, t0, nslots, 1)
        ex.regularity_recombine(t0, basis.regularity_plan(ex, rank)[1],
                                basis.radial_factor(ex, scales[2], basis.k) if basis.k > 0 else None)
    t1 = ex.empty((nc, 2 * nml, Nt, Ng))
    basis.colatitude_plan(ex, Nt, rank).backward(t0.reshape(1, nc * 2 * nml, sb.nl, Ng),
                                                 t1.reshape(1, nc * 2 * nml, Nt, Ng))
    if rank > 0:
        t2 = ex.empty((nc, 2 * nml, Nt * Ng))
        ex.spin_recombine(t1.reshape(nc, 2 * nml, Nt * Ng), t2, basis.recombination_matrix(rank, forward=False))
    else:
        t2 = t1
    P = dist.size
    Ntl = Nt
    if P > 1:
        # [nc][2 nml][Nt][Ng] -> [nc][2 nm][Nt / P][Ng]
        Ntl = dist.theta_range(Nt)[1]
        n_el = nc * 2 * nml * Nt * Ng
        send, recv = ex.empty((n_el,)), ex.empty((n_el,))
        ex.a2a_pack(t2, send, nc * 2 * nml, Nt, 1, Ng, P)
        dist.pcomm.all_to_all(recv, send)
        t2 = ex.empty((nc, 2 * sb.nm, Ntl * Ng))
        ex.a2a_unpack(recv, t2, nc, 1, 2 * nml * P, Ntl * Ng, P)
    g = ex.empty((nc, Np, Ntl, Ng))
    ex.transform(("rfft", Np, sb.Nphi), None, "backward", t2, g, nc, Ntl * Ng)
    return g


def forward(dist, basis, rank, g, scales):
    ex = dist.executor
    sb = basis.sphere
    nc = 3 ** rank
    nml = sb.nml
    Np, Nt, Ng = basis.grid_shape(scales)
    P = dist.size
    Ntl = dist.theta_range(Nt)[1] if P > 1 else Nt
    t1 = ex.empty((nc, 2 * sb.nm, Ntl * Ng))
    ex.transform(("rfft", Np, sb.Nphi), None, "forward", g, t1, nc, Ntl * Ng)
    if P > 1:
        # [nc][2 nm][Nt / P][Ng] -> [nc][2 nml][Nt][Ng]
        n_el = nc * 2 * sb.nm * Ntl * Ng
        send, recv = ex.empty((n_el,)), ex.empty((n_el,))
        ex.a2a_pack(t1, send, nc, 2 * sb.nm, Ntl, Ng, P)
        dist.pcomm.all_to_all(recv, send)
        t1 = ex.empty((nc, 2 * nml, Nt * Ng))
        ex.a2a_unpack(recv, t1, nc, 2 * nml, Ntl * P, Ng, P)
    if rank > 0:
        t2 = ex.empty((nc, 2 * nml, Nt * Ng))
        ex.spin_recombine(t1, t2, basis.recombination_matrix(rank, forward=True))
    else:
        t2 = t1
    t3 = ex.zeros((nc, 2 * nml, sb.nl, Ng))
    basis.colatitude_plan(ex, Nt, rank).forward(t2.reshape(1, nc * 2 * nml, Nt, Ng),
                                                t3.reshape(1, nc * 2 * nml, sb.nl, Ng))
    if isinstance(basis, SurfaceBasis):
        return t3
    ex.regularity_recombine(t3, basis.regularity_plan(ex, rank)[0],
                            basis.radial_factor(ex, scales[2], -basis.k) if basis.k > 0 else None)
    gemm = _radial_gemm(dist, basis, scales[2], nc, True)
    if gemm is not None:
        full = ex.empty((nc, 2 * nml, sb.nl, Ng))
        gemm.apply(t3, full)
        return _unpadded(ex, full, basis.Nr)
    c = ex.empty((nc, 2 * nml, sb.nl, basis.Nr))
    ex.transform(basis.radial.plan_spec(scales[2]), basis.radial, "forward", t3, c, nc * 2 * nml * sb.nl, 1)
    return c



# ==================================================================================================
# ell-dependent radial term lists and operands
# ==================================================================================================

class NonlinearError(ValueError):
    pass


class EllTermList:
    """Linear map between shell coefficient arrays: out[co][i1][ell][:] = sum A[ell] in[ci][i1][ell][:], matrices
    stored as [nl][Nr][Nr] (sphere-surface operands use the leading 1 x Nr / Nr x 1 / 1 x 1 block)."""

    def __init__(self, nco, nci, terms=None):
        self.nco, self.nci = nco, nci
        self.terms = list(terms or [])          # (co, ci, mats [nl][Nr][Nr])

    @staticmethod
    def identity(nc, nl, Nr, nr):
        m = np.zeros((nl, Nr, Nr))
        m[:, np.arange(nr), np.arange(nr)] = 1.0
        return EllTermList(nc, nc, [(c, c, m.copy()) for c in range(nc)])

    def scaled(self, a):
        return EllTermList(self.nco, self.nci, [(co, ci, a * m) for (co, ci, m) in self.terms])

    def __add__(self, other):
        assert (self.nco, self.nci) == (other.nco, other.nci)
        return EllTermList(self.nco, self.nci, self.terms + other.terms).merged()

    def merged(self):
        acc = {}
        for (co, ci, m) in self.terms:
            acc[(co, ci)] = acc[(co, ci)] + m if (co, ci) in acc else np.array(m, dtype=np.float64)
        return EllTermList(self.nco, self.nci, [(co, ci, m) for (co, ci), m in sorted(acc.items()) if np.any(m != 0)])

    def compose(self, inner):
        assert self.nci == inner.nco
        out = []
        for (co, cm, A) in self.terms:
            for (cm2, ci, B) in inner.terms:
                if cm2 == cm:
                    out.append((co, ci, np.matmul(A, B)))
        return EllTermList(self.nco, inner.nci, out).merged()

    def embed(self, row0, col0, nrows, ncols):
        return EllTermList(nrows, ncols, [(co + row0, ci + col0, m) for (co, ci, m) in self.terms])


REG = (-1, +1, 0)


def reg_indices(rank):
    return list(np.ndindex(*((3,) * rank)))


def regtotal(idx):
    return sum(REG[a] for a in idx)


def regularity_allowed(ell, idx):
    """RegularityBasis.regularity_allowed (core/basis.py:3531-3535, vectorised form :3585-3593)."""
    walk = ell
    for a in idx[::-1]:
        dr = REG[a]
        walk = walk + dr
        if walk < 0 or (dr == 0 and walk == 0):
            return False
    return True


def spin_allowed(ell, idx):
    return ell >= abs(sum(REG[a] for a in idx))


def shell_op_termlist(kind, shell, rank_in, k, **kw):
    """Term list of one operator of the reference's SphericalEllOperator family acting on a rank-`rank_in` tensor
    in the k-th shell basis: lap (core/operators.py:4109-4152), grad (:3240-3288), div (:3546-3600), convert
    (ConvertSpherical3D, core/basis.py:4822-4842), lift (LiftShell :5155-5199), interp (ShellRadialInterpolate
    :5823-5889)."""
    from ..tools import shellops as so
    sb = shell.sphere
    nl, Nr, radii, alpha = sb.nl, shell.Nr, shell.radii, shell.alpha
    idx_in = reg_indices(rank_in)

    def stack(fn, allow_in, allow_out):
        m = np.zeros((nl, Nr, Nr))
        for ell in range(nl):
            if allow_in(ell) and allow_out(ell):
                a = np.asarray(fn(ell))
                m[ell, :a.shape[0], :a.shape[1]] = a
        return m

    terms = []
    if kind == "lap":
        for ci, t in enumerate(idx_in):
            rt = regtotal(t)
            terms.append((ci, ci, stack(lambda ell: so.operator_matrix("L", ell, rt, Nr, k, radii, alpha),
                                        lambda ell: regularity_allowed(ell, t), lambda ell: True)))
        return EllTermList(len(idx_in), len(idx_in), terms)
    if kind == "convert":
        dk = kw["dk"]
        E = so.E_power(Nr, k, dk, radii, alpha)
        for ci, t in enumerate(idx_in):
            terms.append((ci, ci, stack(lambda ell: E, lambda ell: regularity_allowed(ell, t), lambda ell: True)))
        return EllTermList(len(idx_in), len(idx_in), terms)
    if kind == "grad":
        idx_out = reg_indices(rank_in + 1)
        for ci, t in enumerate(idx_in):
            rt = regtotal(t)
            for a, (mu, name) in enumerate(((-1, "D-"), (+1, "D+"))):
                to = (a,) + tuple(t)
                co = idx_out.index(to)
                terms.append((co, ci, stack(lambda ell: so.xi(mu, ell + rt) * so.operator_matrix(name, ell, rt, Nr, k, radii, alpha),
                                            lambda ell: regularity_allowed(ell, t), lambda ell: regularity_allowed(ell, to))))
        return EllTermList(len(idx_out), len(idx_in), terms)
    if kind == "div":
        idx_out = reg_indices(rank_in - 1)
        for ci, t in enumerate(idx_in):
            if t[0] == 2:
                continue
            rt = regtotal(t)
            to = tuple(t[1:])
            co = idx_out.index(to)
            if t[0] == 0:
                fn = lambda ell: so.xi(-1, ell + rt + 1) * so.operator_matrix("D+", ell, rt, Nr, k, radii, alpha)
            else:
                fn = lambda ell: so.xi(+1, ell + rt - 1) * so.operator_matrix("D-", ell, rt, Nr, k, radii, alpha)
            terms.append((co, ci, stack(fn, lambda ell: regularity_allowed(ell, t), lambda ell: regularity_allowed(ell, to))))
        return EllTermList(len(idx_out), len(idx_in), terms)
    if kind == "trace":              # SphericalTrace (core/operators.py:1783-1826): Q_out^T trace_spin Q_in, identity in n
        if rank_in != 2:
            raise NotImplementedError("trace of a rank-%d tensor" % rank_in)
        tr = np.zeros(9)
        tr[[1, 3, 8]] = 1.0
        for ci, t in enumerate(idx_in):
            def fn(ell, ci=ci):
                return float(tr @ sph.intertwiner(ell, 2)[:, ci]) * np.eye(Nr)
            m = stack(fn, lambda ell: regularity_allowed(ell, t), lambda ell: True)
            if np.any(m != 0):
                terms.append((0, ci, m))
        return EllTermList(1, len(idx_in), terms)
    if kind == "integ":              # IntegrateShell (core/basis.py:5555-5575): ell = 0 only -> a constant
        from ..tools import jacobi
        z0, w0 = jacobi.quadrature(2 * Nr, 0, 0)
        z0, w0 = np.asarray(z0, dtype=np.float64), np.asarray(w0, dtype=np.float64)
        dR = radii[1] - radii[0]
        r0 = dR / 2 * (z0 + (radii[1] + radii[0]) / dR)
        Qk = np.asarray(jacobi.polynomials(Nr, alpha[0] + k, alpha[1] + k, z0))
        row = (r0 ** 2 * w0 * (r0 / dR) ** (-k)) @ Qk.T * (dR / 2) * (4 * np.pi / np.sqrt(2))
        m = np.zeros((nl, Nr, Nr))
        m[0, 0, :] = row
        return EllTermList(1, 1, [(0, 0, m)])
    if kind == "convert_const":      # ConvertConstantShell (core/basis.py:4782-4819): constant -> ell = 0, k-th basis
        from ..tools import jacobi
        cmv = float(np.asarray(jacobi.polynomials(1, alpha[0], alpha[1], np.array([0.0])))[0, 0]) / np.sqrt(2)
        col = so.E_power(Nr, 0, k, radii, alpha)[:, 0] / cmv
        m = np.zeros((nl, Nr, Nr))
        m[0, :, 0] = col
        return EllTermList(1, 1, [(0, 0, m)])
    if kind == "lift":               # sphere-surface spin components -> regularity components, radial mode n
        n = kw["n"]
        n_idx = n if n >= 0 else Nr + n
        for co, to in enumerate(idx_in):
            for ci, ti in enumerate(idx_in):
                def fn(ell):
                    q = sph.intertwiner(ell, rank_in)[ci, co] if rank_in else 1.0      # Q^T[co, ci]
                    a = np.zeros((Nr, 1))
                    a[n_idx, 0] = q
                    return a
                m = stack(fn, lambda ell: spin_allowed(ell, ti), lambda ell: regularity_allowed(ell, to))
                if np.any(m != 0):
                    terms.append((co, ci, m))
        return EllTermList(len(idx_in), len(idx_in), terms)
    if kind == "interp":             # regularity components -> spin components on the sphere r = position
        vec = so.interpolation(kw["position"], Nr, k, radii, alpha)
        for co, to in enumerate(idx_in):
            for ci, ti in enumerate(idx_in):
                def fn(ell):
                    q = sph.intertwiner(ell, rank_in)[co, ci] if rank_in else 1.0      # Q[co, ci]
                    return q * vec.reshape(1, Nr)
                m = stack(fn, lambda ell: regularity_allowed(ell, ti), lambda ell: spin_allowed(ell, to))
                if np.any(m != 0):
                    terms.append((co, ci, m))
        return EllTermList(len(idx_in), len(idx_in), terms)
    raise ValueError(kind)


def operate_slot_sequences(sb):
    """(extra sequences, slot map [2 nm][nl]) for operator EVALUATION: matrix index ell for a slot covered by its own
    ell_maps box only, nl + j for a slot covered by the boxes listed in seqs[j]; -1 where there is no mode."""
    key = "operate_slots"
    if key not in sb._plans:
        rows_p = sb.packed_ell_rows()
        ps = sb.packed_shape()
        cover = [[[] for _ in range(ps[1])] for _ in range(ps[0])]
        for (ell, i0, i1, j0, j1) in rows_p:
            for i in range(i0, i1):
                for j in range(j0, j1):
                    cover[i][j].append(int(ell))
        rows, cols, ok = sb.pack_index()
        slot = -np.ones((2 * sb.nm, sb.nl), dtype=np.int32)
        seqs, index = [], {}
        for i in range(ps[0]):
            for j in range(ps[1]):
                if not ok[i, j]:
                    continue
                r, l = int(rows[i, j]), int(cols[i, j])
                seq = tuple(cover[i][j])
                if seq == (l,):
                    slot[r, l] = l
                else:
                    if seq not in index:
                        index[seq] = len(seqs)
                        seqs.append(seq)
                    slot[r, l] = sb.nl + index[seq]
        sb._plans[key] = (seqs, slot)
    return sb._plans[key]


class ShOperand:
    """Expression node in spherical coordinates.  basis: ShellBasis (k), SurfaceBasis or None."""
    __array_priority__ = 100.0

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        """numpy scalars and ufuncs applied to operands (np.sqrt(u@u), np.float64(2) * u, ...)"""
        if method != "__call__" or kw:
            return NotImplemented
        if len(inputs) == 1:
            return ShUnary(ufunc, inputs[0])
        a, b = inputs
        if ufunc is np.multiply:
            return a * b if isinstance(a, ShOperand) else b * a
        if ufunc is np.add:
            return a + b if isinstance(a, ShOperand) else b + a
        if ufunc is np.subtract:
            return a - b if isinstance(a, ShOperand) else (-1 * b) + a
        if ufunc is np.true_divide and isinstance(b, numbers.Number):
            return a * (1.0 / b)
        if ufunc is np.matmul:
            return ShProduct(a, b, contract=True)
        return NotImplemented

    @property
    def ncomp(self):
        return 3 ** self.rank

    def __add__(self, other):
        return ShAdd.make(self, other)

    __radd__ = __add__

    def __sub__(self, other):
        return ShAdd.make(self, -1 * other if isinstance(other, ShOperand) else -other)

    def __rsub__(self, other):
        return ShAdd.make(-1 * self, other)

    def __neg__(self):
        return ShScale(-1.0, self)

    def __mul__(self, other):
        if isinstance(other, numbers.Number):
            return ShScale(other, self)
        if isinstance(other, ShOperand):
            return ShProduct(self, other)
        return NotImplemented

    def __rmul__(self, other):
        if isinstance(other, numbers.Number):
            return ShScale(other, self)
        return NotImplemented

    def __matmul__(self, other):
        return ShProduct(self, other, contract=True)

    def eval_g(self):
        return _eval_grid(self)

    def __truediv__(self, other):
        if isinstance(other, numbers.Number):
            return ShScale(1.0 / other, self)
        return NotImplemented

    def __call__(self, **kw):
        """T(r=Ri): interpolation along the radius (core/operators.py interpolate dispatch)."""
        if len(kw) != 1:
            raise ValueError("one coordinate at a time")
        (name, pos), = kw.items()
        if name == self.dist.coordsys.coords[0].name:
            return ShAzimuthalInterp(self, float(pos))
        if name != self.dist.coordsys.radius.name:
            return ShUnsupported("interpolation along %r" % name, self)
        return ShLinear("interp", self, position=float(pos))

    def has_dt(self):
        return any(a.has_dt() for a in getattr(self, "args", ()) if isinstance(a, ShOperand))

    def grid_native(self):
        """Grid data when the operand is formed in grid space (products), else None."""
        return None

    def evaluate(self):
        f = ShellField(self.dist, self.basis, rank=self.rank)
        f._set_device_coeff(self.eval_c())
        return f


class ShAzimuthalInterp(ShOperand):
    """f(phi=phi0) (core/operators.py interpolate dispatch -> SphereBasis azimuthal interpolation): an ANALYSIS-ONLY
    operand (output tasks such as the example's meridional flux slices).  The operand is evaluated on the grid at the
    requested scales and its trigonometric interpolant along phi is evaluated on the host; the result keeps a phi
    axis of size one (a constant axis in the output file)."""

    const_axes = (0,)

    def __init__(self, arg, position):
        self.args, self.position = (arg,), position
        self.dist, self.basis, self.rank = arg.dist, arg.basis, arg.rank
        self.scales = (1.0, 1.0, 1.0)
        self._field = None

    def evaluate(self):
        out = ShAzimuthalInterp(self.args[0], self.position)
        arg = self.args[0]
        out._field = arg if isinstance(arg, ShellField) else arg.evaluate()
        return out

    def require_coeff_space(self):
        self._field.require_coeff_space()

    def change_scales(self, scales):
        self.scales = self._field._remedy(scales)

    def __getitem__(self, layout):
        if layout not in ("g", "grid"):
            raise NotImplementedError("coefficient data of an azimuthal interpolation")
        f = self._field
        f.change_scales(self.scales)
        g = np.asarray(f["g"])
        ax = self.rank
        Np = g.shape[ax]
        c = np.fft.rfft(g, axis=ax) / Np
        k = np.arange(c.shape[ax])
        w = np.where((k == 0) | ((Np % 2 == 0) & (k == Np // 2)), 1.0, 2.0) * np.exp(1j * k * self.position)
        shape = [1] * g.ndim
        shape[ax] = k.size
        return np.sum((c * w.reshape(shape)).real, axis=ax, keepdims=True)

    def eval_c(self):
        raise NotImplementedError("azimuthal interpolation is an output task, not a term of an equation")

    def lin(self, variables):
        raise NonlinearError("azimuthal interpolation in an equation")


class ShUnsupported(ShOperand):
    """Placeholder for expressions that scripts build for output tasks but that nothing evaluates here."""

    def __init__(self, what, arg):
        self.what, self.args = what, (arg,)
        self.dist, self.basis, self.rank = arg.dist, arg.basis, arg.rank

    def eval_c(self):
        raise NotImplementedError(self.what)

    def lin(self, variables):
        raise NonlinearError(self.what)


class ShUnary(ShOperand):
    """ufunc(operand) point by point on the dealiased grid (UnaryGridFunction, core/operators.py:460-560); an
    analysis-only path: the grid data make a round trip through the host."""

    def __init__(self, func, arg):
        if arg.rank:
            raise NotImplementedError("ufuncs of tensor fields")
        self.func, self.args = func, (arg,)
        self.dist, self.basis, self.rank = arg.dist, arg.basis, 0

    def grid_native(self):
        ex = self.dist.executor
        return ex.from_host(np.ascontiguousarray(self.func(np.asarray(ex.download(self.args[0].eval_g())))))

    def eval_g(self):
        return self.grid_native()

    def eval_c(self):
        return forward(self.dist, self.basis, 0, self.grid_native(), self.basis.dealias)

    def lin(self, variables):
        raise NonlinearError("grid functions are nonlinear")


class ShScale(ShOperand):
    def __init__(self, a, arg):
        self.a, self.arg, self.args = float(a), arg, (arg,)
        self.dist, self.basis, self.rank = arg.dist, arg.basis, arg.rank

    def grid_native(self):
        g = self.arg.grid_native()
        if g is None:
            return None
        ex = self.dist.executor
        out = ex.empty(tuple(g.shape))
        ex.lincomb(out, [g], [self.a])
        return out

    def eval_c(self):
        ex = self.dist.executor
        c = self.arg.eval_c()
        out = ex.empty(tuple(c.shape))
        ex.lincomb(out, [c], [self.a])
        return out

    def lin(self, variables):
        d, dt = self.arg.lin(variables)
        return {i: t.scaled(self.a) for i, t in d.items()}, dt


def _common_basis(a, b):
    """Sum basis: shells of the same geometry meet in the larger k (core/basis.py ShellBasis.__add__ :4415-4424)."""
    if isinstance(a.basis, ShellBasis) and isinstance(b.basis, ShellBasis):
        return a.basis if a.basis.k >= b.basis.k else b.basis
    if a.basis is None and isinstance(b.basis, ShellBasis):
        return b.basis
    if b.basis is None and isinstance(a.basis, ShellBasis):
        return a.basis
    if a.basis == b.basis:
        return a.basis
    raise NotImplementedError("sum of operands on different bases")


def _converted(x, basis):
    if isinstance(basis, ShellBasis) and x.basis is None:
        return ShLinear("convert_const", x, basis=basis)
    if isinstance(basis, ShellBasis) and x.basis.k != basis.k:
        return ShLinear("convert", x, dk=basis.k - x.basis.k)
    return x


class ShAdd(ShOperand):
    @staticmethod
    def make(a, b):
        for x, y in ((a, b), (b, a)):
            if isinstance(x, numbers.Number):
                if x == 0:
                    return y
                raise NotImplementedError("adding a number to a shell field")
        return ShAdd(a, b)

    def __init__(self, a, b):
        if a.rank != b.rank:
            raise ValueError("cannot add tensors of different rank")
        self.dist, self.rank = a.dist, a.rank
        self.basis = _common_basis(a, b)
        self.args = (_converted(a, self.basis), _converted(b, self.basis))

    def eval_c(self):
        ex = self.dist.executor
        cs = [x.eval_c() for x in self.args]
        out = ex.empty(tuple(cs[0].shape))
        ex.lincomb(out, cs, [1.0, 1.0])
        return out

    def lin(self, variables):
        out, anydt = {}, None
        for x in self.args:
            d, dt = x.lin(variables)
            if anydt is None:
                anydt = dt
            elif anydt != dt:
                raise NonlinearError("dt and non-dt terms inside one sum node")
            for i, t in d.items():
                out[i] = out[i] + t if i in out else t
        return out, bool(anydt)


class ShLinear(ShOperand):
    """lap / grad / div / convert / lift / interp of an operand."""

    def __init__(self, kind, arg, **kw):
        if not isinstance(arg, ShOperand):
            raise ValueError("%s needs a field operand" % kind)
        self.kind, self.arg, self.args, self.kw = kind, arg, (arg,), kw
        self.dist = arg.dist
        ab = arg.basis
        if kind in ("lap", "grad", "div", "convert", "interp", "trace", "integ") and not isinstance(ab, ShellBasis):
            raise NotImplementedError("%s of an operand without a shell basis" % kind)
        if kind == "lap":
            self.basis, self.rank = ab.derivative_basis(2), arg.rank
        elif kind == "grad":
            self.basis, self.rank = ab.derivative_basis(1), arg.rank + 1
        elif kind == "div":
            if arg.rank < 1:
                raise ValueError("div needs a tensor of rank >= 1")
            self.basis, self.rank = ab.derivative_basis(1), arg.rank - 1
        elif kind == "convert":
            self.basis, self.rank = ab.derivative_basis(kw["dk"]), arg.rank
        elif kind == "interp":
            self.basis, self.rank = ab.S2_basis(kw["position"]), arg.rank
        elif kind == "trace":
            self.basis, self.rank = ab, arg.rank - 2
        elif kind == "integ":
            if arg.rank:
                raise NotImplementedError("integ of a tensor")
            self.basis, self.rank = None, 0
        elif kind == "convert_const":
            self.basis, self.rank = kw["basis"], 0
        elif kind == "lift":
            if not isinstance(ab, SurfaceBasis):
                raise NotImplementedError("Lift of an operand that is not a surface field")
            self.basis, self.rank = kw["basis"], arg.rank
        else:
            raise ValueError(kind)
        self._dev = None

    @property
    def shell(self):
        if self.kind in ("lift", "convert_const"):
            return self.kw["basis"]
        b = self.arg.basis
        return b.shell if isinstance(b, SurfaceBasis) else b

    def termlist(self):
        kw = {k: v for k, v in self.kw.items() if k != "basis"}
        k = self.kw["basis"].k if self.kind in ("lift", "convert_const") else self.arg.basis.k
        return shell_op_termlist(self.kind, self.shell, self.arg.rank, k, **kw)

    def eval_c(self):
        ex = self.dist.executor
        shell = self.shell
        sb = shell.sphere
        if self.kind == "convert":
            # Convert.operate (core/operators.py:1627-1638): an argument that sits in grid space is copied and then
            # transformed straight into the output basis; only coefficient-space arguments see the E matrices
            g = self.arg.grid_native()
            if g is not None:
                return forward(self.dist, self.basis, self.rank, g, self.basis.dealias)
        if self._dev is None or self._dev[0] is not ex:
            # evaluation follows SphericalEllOperator.operate (core/operators.py:3132-3160), which loops over the
            # ell_maps bounding boxes and ACCUMULATES: a slot covered by several boxes receives the sum of their
            # matrices (the per-group subproblem matrices used by the solvers do not have this overlap)
            tl = self.termlist()
            seqs, slot_map = operate_slot_sequences(sb)
            terms = [(co, ci, np.concatenate([m] + [sum(m[l] for l in seq)[None] for seq in seqs])) for (co, ci, m) in tl.terms]
            slot_map = np.ascontiguousarray(slot_map[2 * sb.m0:2 * (sb.m0 + sb.nml)])
            self._dev = (ex, ex.make_ell_terms(sb.nml, sb.nl, shell.Nr, tl.nco, terms, slot_map))
        x = _padded(ex, self.arg.eval_c(), shell.Nr)
        y = ex.empty((self.ncomp, 2 * sb.nml, sb.nl, shell.Nr))
        self._dev[1].apply(x, y)
        return _unpadded(ex, y, self.basis.Nr if self.basis is not None else 1)

    def eval_g(self):
        return _eval_grid(self)

    def lin(self, variables):
        d, dt = self.arg.lin(variables)
        tl = self.termlist()
        return {i: tl.compose(t) for i, t in d.items()}, dt


class ShDt(ShOperand):
    def __init__(self, arg):
        self.arg, self.args = arg, (arg,)
        self.dist, self.basis, self.rank = arg.dist, arg.basis, arg.rank

    def has_dt(self):
        return True

    def lin(self, variables):
        d, dt = self.arg.lin(variables)
        if dt:
            raise NonlinearError("nested time derivatives")
        return d, True

    def eval_c(self):
        raise ValueError("dt() cannot be evaluated")


def _padded(ex, c, Nr):
    """[nc][2 nm][nl][nr] -> [nc][2 nm][nl][Nr] (surface operands have nr = 1)"""
    if int(c.shape[-1]) == Nr:
        return c
    out = ex.zeros(tuple(c.shape[:-1]) + (Nr,))
    ex.assign(out[..., :int(c.shape[-1])], c)
    return out


def _unpadded(ex, c, nr):
    if int(c.shape[-1]) == nr:
        return c
    out = ex.empty(tuple(c.shape[:-1]) + (nr,))
    ex.assign(out, c[..., :nr])
    return out


def _eval_grid(x):
    """grid data [nc][Nphi_g][Ntheta_g][Nr_g] of an operand at the dealias scales"""
    return backward(x.dist, x.basis, x.rank, x.eval_c(), x.basis.dealias)


class ShProduct(ShOperand):
    """a * b (tensor product) or a @ b (contraction of the last index of a with the first of b).  With a radial NCC
    (RadialField) as one factor the product is linear in the other one (LHS terms: rvec*lift(tau), b*er); otherwise
    it is evaluated on the dealiased grid (MultiplyFields / DotProduct, core/arithmetic.py:586-674)."""

    def __init__(self, a, b, contract=False):
        self.args, self.contract = (a, b), contract
        self.dist = a.dist
        if contract and (a.rank < 1 or b.rank < 1):
            raise ValueError("dot product needs tensors of rank >= 1")
        self.rank = a.rank + b.rank - (2 if contract else 0)
        ncc = [x for x in (a, b) if isinstance(x, RadialField)]
        other = [x for x in (a, b) if not isinstance(x, RadialField)]
        if len(ncc) == 1:
            self.basis = other[0].basis              # k_ncc = 0: the operand's basis
        elif not ncc:
            if not (isinstance(a.basis, ShellBasis) and isinstance(b.basis, ShellBasis)):
                raise NotImplementedError("grid products of operands without shell bases")
            self.basis = a.basis.clone_with(k=a.basis.k + b.basis.k)
        else:
            raise NotImplementedError("product of two radial fields")

    def grid_native(self):
        """grid data of the product at the dealias scales (the layout the product is formed in)"""
        a, b = self.args
        ex = self.dist.executor
        Np, Nt, Ng = self.basis.grid_shape(self.basis.dealias)
        Nt = self.dist.theta_range(Nt)[1]
        ga = a.grid_broadcast(ex, self.basis, (Np, Nt, Ng)) if isinstance(a, RadialField) else a.eval_g()
        gb = b.grid_broadcast(ex, self.basis, (Np, Nt, Ng)) if isinstance(b, RadialField) else b.eval_g()
        terms, nout = _bilinear_terms3(a.rank, b.rank, self.contract)
        out = ex.empty((nout, Np, Nt, Ng))
        ex.bilinear(out, nout, ga, gb, Np * Nt * Ng, terms)
        return out

    def eval_c(self):
        return forward(self.dist, self.basis, self.rank, self.grid_native(), self.basis.dealias)

    def eval_g(self):
        return self.grid_native()

    def lin(self, variables):
        a, b = self.args
        if isinstance(a, RadialField):
            ncc, arg, ncc_first = a, b, True
        elif isinstance(b, RadialField):
            ncc, arg, ncc_first = b, a, False
        else:
            raise NonlinearError("products of fields are nonlinear")
        if not isinstance(arg.basis, ShellBasis):
            raise NotImplementedError("NCC product with an operand that has no shell basis")
        d, isdt = arg.lin(variables)
        tl = ncc_termlist(ncc, arg.basis, arg.rank, ncc_first, self.contract)
        return {i: tl.compose(t) for i, t in d.items()}, isdt


def _bilinear_terms3(rank_a, rank_b, contract):
    ia_list = list(np.ndindex(*((3,) * rank_a)))
    ib_list = list(np.ndindex(*((3,) * rank_b)))
    out_list = list(np.ndindex(*((3,) * (rank_a + rank_b - (2 if contract else 0)))))
    terms = []
    for ia, ta in enumerate(ia_list):
        for ib, tb in enumerate(ib_list):
            if contract:
                if ta[-1] != tb[0]:
                    continue
                tc = tuple(ta[:-1]) + tuple(tb[1:])
            else:
                tc = tuple(ta) + tuple(tb)
            terms.append((out_list.index(tc), ia, ib, 1.0))
    return terms, len(out_list)


def ncc_termlist(ncc, arg_basis, rank_arg, ncc_first, contract):
    """LHS matrix blocks of a product with a radial NCC (Basis._last_axis_field_ncc_matrix, core/basis.py:283-330, with
    ShellRadialBasis._last_axis_component_ncc_matrix :3879-3910): block (gamma <- beta) at ell =
    sum_alpha Gamma_ell[alpha, beta, gamma] * Mult(a_alpha), Mult = multiplication by the NCC's regularity component
    alpha as a function of r in the operand's Jacobi family (exact Gauss quadrature; the reference's Clenshaw sum
    of the same polynomial)."""
    from ..tools import jacobi
    shell = arg_basis
    sb = shell.sphere
    nl, Nr, k = sb.nl, shell.Nr, arg_basis.k
    a_fam, b_fam = k + shell.alpha[0], k + shell.alpha[1]
    # NCC regularity components as functions of z: coefficients in the k = 0 family, evaluated at the quadrature nodes
    coef = ncc.regularity_coefficients()                     # [3^rank_ncc][Nr]
    nq = Nr + coef.shape[1] + 2
    zq, wq = jacobi.quadrature(nq, a_fam, b_fam)
    zq, wq = np.asarray(zq, dtype=np.float64), np.asarray(wq, dtype=np.float64)
    P = np.asarray(jacobi.polynomials(Nr, a_fam, b_fam, zq))                         # [Nr][nq]
    P0 = np.asarray(jacobi.polynomials(coef.shape[1], shell.alpha[0], shell.alpha[1], zq))
    mult = []
    for alpha_c in range(coef.shape[0]):
        f = coef[alpha_c] @ P0
        if np.any(np.abs(coef[alpha_c]) > 1e-14):
            M = (P * (wq * f)) @ P.T
            M[np.abs(M) < 1e-13 * np.abs(M).max()] = 0.0         # quadrature round-off outside the band of the polynomial
            mult.append(M)
        else:
            mult.append(None)
    rank_ncc = ncc.rank
    rank_out = rank_ncc + rank_arg - (2 if contract else 0)
    idx_arg, idx_out = reg_indices(rank_arg), reg_indices(rank_out)
    acc = {}
    for ell in range(nl):
        if ncc_first:
            G = gamma_regularity(rank_ncc, rank_arg, 0, ell, ell, contract)          # [alpha, beta, gamma]
        else:
            G = gamma_regularity(rank_arg, rank_ncc, ell, 0, ell, contract).transpose(1, 0, 2)
        if np.abs(G.imag).max() > 1e-12:
            raise NotImplementedError("complex product coefficients for real fields")
        G = G.real
        for be, tb in enumerate(idx_arg):
            if not regularity_allowed(ell, tb):
                continue
            for ga, tg in enumerate(idx_out):
                if not regularity_allowed(ell, tg):
                    continue
                for al in range(G.shape[0]):
                    if mult[al] is None or abs(G[al, be, ga]) <= 1e-6:             # ncc_cutoff of the reference
                        continue
                    m = acc.setdefault((ga, be), np.zeros((nl, Nr, Nr)))
                    m[ell] += G[al, be, ga] * mult[al]
    return EllTermList(len(idx_out), len(idx_arg), [(co, ci, m) for (co, ci), m in sorted(acc.items())])


def lap(a):
    return ShLinear("lap", a)


def trace(a):
    return ShLinear("trace", a)


def integ(a, *coords):
    return ShLinear("integ", a)


def grad(a):
    return ShLinear("grad", a)


def div(a):
    return ShLinear("div", a)


def dt(a):
    return ShDt(a)


def Lift(a, basis, n):
    return ShLinear("lift", a, basis=basis, n=int(n))


class ConstField(ShOperand):
    """A field without bases in a SphericalCoordinates distributor (tau_p): one number."""

    def __init__(self, dist, name=None):
        self.dist, self.name, self.basis, self.rank = dist, name, None, 0
        self._value = np.zeros((1, 1, 1))
        self._host_dirty = True          # the host copy has to reach the solver's state vector before the next step
        self._pull = None                # set by the solver after a step: fetches the solved value when somebody asks
        self.args = ()

    def __repr__(self):
        return self.name or "<ConstField %d>" % id(self)

    @property
    def value(self):
        """The number, as a (1, 1, 1) array.  While a solver steps, the solved value stays on the device (no host round
        trip per step); it is fetched here, on demand -- and since the caller may write into the array it gets, the
        host copy counts as modified from then on (one small upload before the next step)."""
        if self._pull is not None:
            pull, self._pull = self._pull, None
            self._value[...] = pull()
        self._host_dirty = True
        return self._value

    def __getitem__(self, key):
        return self.value

    def __setitem__(self, key, data):
        self._pull = None
        self._value[...] = data
        self._host_dirty = True

    def change_scales(self, scales):
        pass

    def has_dt(self):
        return False

    def lin(self, variables):
        for i, v in enumerate(variables):
            if v is self:
                shell = [x.shell for x in variables if hasattr(x, "shell")][0]
                sb = shell.sphere
                return {i: EllTermList.identity(1, sb.nl, getattr(shell, "_root", shell).Nr, 1)}, False
        raise NonlinearError("%r is not a problem variable" % (self,))

    def eval_c(self):
        raise NotImplementedError("constants are converted to a shell basis before evaluation")

    # solver interface
    def _cshape_in(self, shell):
        sb = shell.sphere
        return (1, 2 * sb.nml, sb.nl, 1)


class RadialField(ShOperand):
    """Tensor field that depends on r only (NCC: er, rvec): coordinate components on the radial Gauss grid, kept on the
    host (it only enters the LHS matrices)."""

    def __init__(self, dist, shell, rank=0, name=None):
        self.dist, self.shell_basis, self.rank, self.name = dist, shell, rank, name
        self.basis = shell
        self.args = ()
        self._g = np.zeros((3,) * rank + (1, 1, shell.Nr))

    def __getitem__(self, key):
        if key not in ("g", "grid"):
            raise NotImplementedError("coefficient access of radial fields")
        return self._g

    def __setitem__(self, key, data):
        self._g[...] = data

    def has_dt(self):
        return False

    def lin(self, variables):
        raise NonlinearError("a radial field is a coefficient, not a variable")

    def grid_broadcast(self, ex, basis, shape):
        """coordinate components on the dealiased grid [3^rank][Nphi_g][Ntheta_local][Nr_g] (analysis products such as
        er @ flux): the radial profile is re-sampled spectrally on the finer radial grid and repeated over the angles"""
        from ..tools import jacobi
        shell = self.shell_basis
        Np, Nt, Ng = shape
        key = ("ncc_grid", id(ex), id(self), shape)
        store = getattr(shell, "_root", shell)._plans
        if key not in store or store[key][0] != self._g.tobytes():
            nc = 3 ** self.rank
            z, w = jacobi.quadrature(shell.Nr, shell.alpha[0], shell.alpha[1])
            P = np.asarray(jacobi.polynomials(shell.Nr, shell.alpha[0], shell.alpha[1], z))
            coef = self._g.reshape(nc, shell.Nr) @ (P * np.asarray(w, dtype=np.float64)).T
            zg, _ = jacobi.quadrature(Ng, shell.alpha[0], shell.alpha[1])
            Pg = np.asarray(jacobi.polynomials(shell.Nr, shell.alpha[0], shell.alpha[1], zg), dtype=np.float64)
            prof = coef @ Pg                                   # [nc][Ng]
            full = np.ascontiguousarray(np.broadcast_to(prof[:, None, None, :], (nc, Np, Nt, Ng)))
            store[key] = (self._g.tobytes(), ex.from_host(full))
        return store[key][1]

    def regularity_coefficients(self):
        """[3^rank][Nr]: Jacobi (k = 0 family) coefficients of the regularity components at ell = 0."""
        from ..tools import jacobi
        shell = self.shell_basis
        nc = 3 ** self.rank
        g = self._g.reshape(nc, shell.Nr)
        if self.rank:
            U = SphericalCoordinates.U_forward(self.rank)
            spin = U @ g
            if np.abs(spin.imag).max() > 1e-14:
                raise NotImplementedError("radial NCC with angular components")
            reg = sph.intertwiner(0, self.rank).T @ spin.real
        else:
            reg = g
        z, w = jacobi.quadrature(shell.Nr, shell.alpha[0], shell.alpha[1])
        P = np.asarray(jacobi.polynomials(shell.Nr, shell.alpha[0], shell.alpha[1], z))
        return reg @ (P * np.asarray(w, dtype=np.float64)).T


class ShellField(ShOperand):
    """Tensor field on a ShellBasis (or one of its surfaces): device-resident data with lazily synchronised host mirrors, `['g']` / `['c']`
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
        self.args = ()

    @property
    def shell(self):
        return self.basis.shell if isinstance(self.basis, SurfaceBasis) else self.basis

    @property
    def k(self):
        return self.basis.k

    @property
    def tensorsig(self):
        return (self.dist.coordsys,) * self.rank

    # ---- expression protocol ---------------------------------------------------------------------------------------
    def eval_c(self):
        return self.require_coeff_space()

    def lin(self, variables):
        for i, v in enumerate(variables):
            if v is self:
                sb = self.basis.sphere
                return {i: EllTermList.identity(self.ncomp, sb.nl, self.shell.Nr, self.basis.Nr)}, False
        raise NonlinearError("%r is not a problem variable" % (self,))

    def has_dt(self):
        return False

    def evaluate(self):
        return self

    def __repr__(self):
        return self.name or "<ShellField %d>" % id(self)

    @property
    def ex(self):
        return self.dist.executor

    def _cshape(self):
        sb = self.basis.sphere
        return (self.ncomp, 2 * sb.nml, sb.nl, self.basis.Nr)

    def _set_device_coeff(self, c):
        self._c = c
        self.layout = "c"
        self._authority = "device"
        self._g = None

    def _global_shape(self, layout, scales):
        t = (3,) * self.rank
        if layout == "g":
            return t + self.basis.grid_shape(scales)
        return t + self.basis.sphere.packed_shape() + (self.basis.Nr,)

    def _local_slices(self, layout, scales):
        """slices of the global user array held by this rank: colatitude block in grid space, block of the packed
        azimuthal axis in coefficient space (Layout.local_chunks of the reference, core/distributor.py:357-385)"""
        sl = [slice(None)] * (self.rank + 3)
        d = self.dist
        if d.size > 1:
            if layout == "g":
                t0, nt = d.theta_range(self.basis.grid_shape(scales)[1])
                sl[self.rank + 1] = slice(t0, t0 + nt)
            else:
                n = self.basis.sphere.packed_shape()[0]
                if n % (2 * d.size):
                    raise ValueError("packed azimuthal axis of %d does not divide over %d ranks" % (n, d.size))
                sl[self.rank] = slice(d.rank * (n // d.size), (d.rank + 1) * (n // d.size))
        return tuple(sl)

    def _user_shape(self, layout, scales):
        g = self._global_shape(layout, scales)
        return tuple(len(range(*sl.indices(n))) for sl, n in zip(self._local_slices(layout, scales), g))

    def _natural_from_packed(self, h):
        """user (packed, local block) coefficients -> natural [ncomp][2 nml][nl][Nr] of the local wavenumbers.  On several
        ranks the folded part of the packed layout belongs to other ranks' wavenumbers: the blocks are all-gathered
        (user access is a synchronisation point, not part of the step)."""
        sb = self.basis.sphere
        rows, cols, ok = sb.pack_index()
        h = h.reshape((self.ncomp, -1, rows.shape[1], self.basis.Nr))
        if self.dist.size > 1:
            h = self.dist.pcomm.all_gather_host(h, axis=1)
        nat = np.zeros((self.ncomp, 2 * sb.nm, sb.nl, self.basis.Nr))
        for c in range(self.ncomp):
            nat[c][rows[ok], cols[ok], :] = h[c][ok]
        return np.ascontiguousarray(nat[:, 2 * sb.m0:2 * (sb.m0 + sb.nml)])

    def _packed_from_natural(self, nat):
        sb = self.basis.sphere
        if self.dist.size > 1:
            nat = self.dist.pcomm.all_gather_host(np.ascontiguousarray(nat), axis=1)
        rows, cols, ok = sb.pack_index()
        out = np.zeros((self.ncomp,) + rows.shape + (self.basis.Nr,))
        for c in range(self.ncomp):
            out[c][ok] = nat[c][rows[ok], cols[ok], :]
        out = out.reshape(self._global_shape("c", self.scales))
        return np.ascontiguousarray(out[self._local_slices("c", self.scales)])

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
            nat = self._natural_from_packed(self._host)
            if self._c is None:
                self._c = self.ex.zeros(self._cshape())
            self.ex.upload(self._c, nat)
            self.layout = "c"
        else:
            shape = (self.ncomp,) + self._user_shape("g", sc)[self.rank:]
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
                self._host = self._packed_from_natural(nat).reshape(shape)
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

    def cfl_frequency_max(self):
        """max over the dealiased grid of the advective CFL frequency of this velocity field (Spherical3DAdvectiveCFL,
        core/basis.py:6183-6204), reduced on the device."""
        if self.rank != 1 or not isinstance(self.basis, ShellBasis):
            raise ValueError("CFL velocity must be a vector field on a shell")
        ex = self.ex
        key = ("cfl", id(ex))
        shell = self.basis
        store = getattr(shell, "_root", shell)._plans
        if key not in store:
            sc = shell.dealias
            r = shell.radius_grid(sc[2])
            Lmax = shell.sphere.Lmax
            h = r / np.sqrt(Lmax * (Lmax + 1)) if Lmax > 0 else np.full_like(r, np.inf)
            dr = np.gradient(r, edge_order=2) * sc[2]
            store[key] = (ex.from_host(np.ascontiguousarray(1.0 / h)), ex.from_host(np.ascontiguousarray(1.0 / np.abs(dr))))
        inv_h, inv_dr = store[key]
        g = self.eval_g()
        return ex.cfl_max_spherical(g, inv_h, inv_dr)

    def fill_random(self, layout=None, scales=None, seed=None, chunk_size=2 ** 20, distribution="standard_normal", **kw):
        """The reference's reproducible global random stream (core/field.py:898-943, tools/random_arrays.py:7-55)."""
        if scales is not None:
            self.change_scales(scales)
        layout = "c" if (layout or self.layout) in ("c", "coeff") else "g"
        shape = self._global_shape(layout, self.scales)
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
        self[layout] = out.reshape(shape)[self._local_slices(layout, self.scales)]


# ==================================================================================================
# problems and solvers (per-ell systems)
# ==================================================================================================

class ShellProblem:
    def __init__(self, variables, namespace=None, time="t"):
        self.variables = list(variables)
        self.dist = self.variables[0].dist
        shells = [v.shell for v in self.variables if hasattr(v, "shell")]
        self.shell = getattr(shells[0], "_root", shells[0])
        self.equations = []
        self.namespace = dict(lap=lap, grad=grad, div=div, dt=dt, Lift=Lift, lift=Lift, trace=trace, integ=integ,
                              Laplacian=lap, Gradient=grad, Divergence=div, TimeDerivative=dt, Trace=trace,
                              Integrate=integ, np=np, numpy=np)
        if namespace:
            self.namespace.update(namespace)
        for v in self.variables:
            if v.name:
                self.namespace[v.name] = v

    def _parse(self, side):
        if isinstance(side, (ShOperand, numbers.Number)):
            return side
        return eval(side, dict(self.namespace))

    def add_equation(self, equation, condition=None):
        from .problems import _split_equation
        if isinstance(equation, str):
            lhs_s, rhs_s = _split_equation(equation)
            lhs, rhs = self._parse(lhs_s), self._parse(rhs_s)
        else:
            lhs, rhs = [self._parse(x) for x in equation]
        if not isinstance(lhs, ShOperand):
            raise ValueError("LHS must involve the problem variables")
        if isinstance(rhs, ShOperand) and rhs.has_dt():
            raise ValueError("time derivatives must be on the LHS")
        M, L = self._linearize(lhs)
        if isinstance(rhs, numbers.Number):
            F = None if rhs == 0 else float(rhs)
        else:
            if rhs.rank != lhs.rank:
                raise ValueError("LHS and RHS tensor signatures differ")
            F = _converted(rhs, lhs.basis) if isinstance(lhs.basis, ShellBasis) else rhs
        eq = dict(lhs=lhs, basis=lhs.basis, rank=lhs.rank, ncomp=lhs.ncomp, M=M, L=L, F=F,
                  string=equation if isinstance(equation, str) else None)
        self.equations.append(eq)
        return eq

    def _linearize(self, lhs):
        terms = []

        def flatten(node, scale):
            if isinstance(node, ShAdd):
                for a in node.args:
                    flatten(a, scale)
            elif isinstance(node, ShScale):
                flatten(node.arg, scale * node.a)
            else:
                terms.append((scale, node))
        flatten(lhs, 1.0)
        M, L = {}, {}
        for scale, node in terms:
            try:
                d, isdt = node.lin(self.variables)
            except NonlinearError as e:
                raise ValueError("LHS must be linear in the problem variables: %s" % e)
            tgt = M if isdt else L
            for i, tl in d.items():
                tl = tl.scaled(scale)
                tgt[i] = tgt[i] + tl if i in tgt else tl
        return M, L

    def build_solver(self, *args, **kw):
        return self.solver_class(self, *args, **kw)


class ShellIVP(ShellProblem):
    @property
    def solver_class(self):
        return ShellInitialValueSolver


class ShellLBVP(ShellProblem):
    @property
    def solver_class(self):
        return ShellBoundaryValueSolver


class _EllPack:
    def __init__(self):
        self.mats = []

    def add(self, dev):
        self.mats.append(dev)
        return len(self.mats) - 1

    def matvec(self, mid, x, y):
        self.mats[mid].apply(x, y)


def _valid_modes(basis, rank, nl, Nr):
    """[ncomp][nl][Nr] validity of (component, ell, n): regularity components of a shell field, spin components of
    a surface field (n = 0 only)."""
    out = np.zeros((3 ** rank, nl, Nr), dtype=bool)
    if basis is None:                       # constants: the (ell = 0, n = 0) mode only
        out[0, 0, 0] = True
        return out
    for c, idx in enumerate(reg_indices(rank)):
        for ell in range(nl):
            if isinstance(basis, SurfaceBasis):
                out[c, ell, 0] = spin_allowed(ell, idx)
            else:
                out[c, ell, :] = regularity_allowed(ell, idx)
    return out


class ShellSolverBase:
    """Per-ell systems: the matrices depend on ell only (matrix_dependence of the reference), every
    (m, part) slot of that ell is a right-hand-side column.  System vectors are [R][2 nm][nl][Nr]: one system component
    per component of a shell-basis variable / equation, while the components that have a single radial point (surface
    tau fields, constants, boundary rows) are PACKED along n of shared components, so that the system stays a small
    number of full Nr x Nr blocks (dense GEMM blocks for the inverse) instead of many single-row / single-column ones."""

    def __init__(self, problem):
        self.problem, self.dist = problem, problem.dist
        self.ex = self.dist.executor
        self.shell = shell = problem.shell
        self.variables = problem.variables
        sb = shell.sphere
        self.nm, self.nl, self.Nr = sb.nml, sb.nl, shell.Nr          # nm: LOCAL azimuthal wavenumbers
        self.m0 = sb.m0
        self.vmap = self._pack_layout([(v.basis, v.ncomp) for v in self.variables])
        self.emap = self._pack_layout([(eq["basis"], eq["ncomp"]) for eq in problem.equations])
        self.R = max(sc for m in self.vmap for (sc, off, nr) in m) + 1
        Re = max(sc for m in self.emap for (sc, off, nr) in m) + 1
        nv = sum(nr for m in self.vmap for (sc, off, nr) in m)
        ne = sum(nr for m in self.emap for (sc, off, nr) in m)
        if nv != ne or Re != self.R:
            raise ValueError("the problem is not square: %d equation rows for %d unknowns per (m, ell)" % (ne, nv))
        self.nx, self.ny = 2 * self.nm, self.nl * self.Nr          # timestepper buffers: (R, nx, ny) elements
        self.col_valid = self._packed_valid([(v.basis, v.rank) for v in self.variables], self.vmap)
        self.row_valid = self._packed_valid([(eq["basis"], eq["rank"]) for eq in problem.equations], self.emap)
        self.M_tl = self._system_termlist("M")
        self.L_tl = self._system_termlist("L")
        self.pack = _EllPack()
        mk = lambda tl: self.ex.make_ell_terms(self.nm, self.nl, self.Nr, self.R, tl.terms)
        self.M_id = self.pack.add(_Reshaped(self, mk(self.M_tl)))
        self.L_id = self.pack.add(_Reshaped(self, mk(self.L_tl)))
        self.X = self.ex.zeros((self.R, self.nx, self.ny))
        self.X4 = self.X.reshape(self.R, 2 * self.nm, self.nl, self.Nr)

    def _pack_layout(self, items):
        """items: (basis, ncomp) per variable / equation -> per item a list over its components of
        (system component, n offset, radial size)."""
        out, sc = [None] * len(items), 0
        for i, (basis, nc) in enumerate(items):           # full-radius components first
            if isinstance(basis, ShellBasis):
                out[i] = [(sc + c, 0, self.Nr) for c in range(nc)]
                sc += nc
        off = self.Nr                                      # then the single-point ones, packed along n
        for i, (basis, nc) in enumerate(items):
            if out[i] is None:
                lst = []
                for c in range(nc):
                    if off >= self.Nr:                     # open a new packed component
                        sc, off = sc + 1, 0
                    lst.append((sc - 1, off, 1))
                    off += 1
                out[i] = lst
        return out

    def _packed_valid(self, items, maps):
        valid = np.zeros((self.R, self.nl, self.Nr), dtype=bool)
        for (basis, rank), m in zip(items, maps):
            v = _valid_modes(basis, rank, self.nl, self.Nr)
            for c, (sc, off, nr) in enumerate(m):
                valid[sc, :, off:off + nr] = v[c, :, :nr]
        return valid

    def _system_termlist(self, which):
        blocks = {}
        Nr = self.Nr
        for eq, em in zip(self.problem.equations, self.emap):
            for i, t in eq[which].items():
                vm = self.vmap[i]
                for (co, ci, m) in t.terms:
                    (so, oo, nro), (si, oi, nri) = em[co], vm[ci]
                    blk = blocks.setdefault((so, si), np.zeros((self.nl, Nr, Nr)))
                    blk[:, oo:oo + nro, oi:oi + nri] += m[:, :nro, :nri]
        out = []
        for (co, ci), m in sorted(blocks.items()):
            m = m * self.row_valid[co][:, :, None] * self.col_valid[ci][:, None, :]
            m[np.abs(m) < 1e-12] = 0.0          # entry_cutoff of the reference's subproblem matrices (core/subsystems.py:536)
            if np.any(m != 0):
                out.append((co, ci, m))
        return EllTermList(self.R, self.R, out)

    def _dense(self, tl, ell):
        A = np.zeros((self.R * self.Nr, self.R * self.Nr))
        for (co, ci, m) in tl.terms:
            A[co * self.Nr:(co + 1) * self.Nr, ci * self.Nr:(ci + 1) * self.Nr] += m[ell]
        return A

    _dinv = None

    def _inverse_terms(self, a, b, old=None):
        """Per-ell inverse of (a M + b L) on the valid modes, formed and inverted on the device
        (executor.make_dense_inverse: M_ell and L_ell are uploaded once; a change of the timestep costs no host work),
        then applied as a term list of dense blocks."""
        if self._dinv is None:
            ells = range(self.nl)
            Ms = [self._dense(self.M_tl, ell) for ell in ells]
            Ls = [self._dense(self.L_tl, ell) for ell in ells]
            # ell < first local m: no local modes -> everything invalid -> a zero block
            rvs = [self.row_valid[:, ell, :].reshape(-1) & (ell >= self.m0) for ell in ells]
            cvs = [self.col_valid[:, ell, :].reshape(-1) & (ell >= self.m0) for ell in ells]
            for ell, (rv, cv) in enumerate(zip(rvs, cvs)):
                if rv.sum() != cv.sum():
                    raise ValueError("ell = %d: %d valid equation modes for %d valid variable modes" % (ell, rv.sum(), cv.sum()))
            self._dinv = self.ex.make_dense_inverse(Ms, Ls, rvs, cvs, complex_=False)
        flat = self._dinv.compute(a, b)
        inner = old.dev if isinstance(old, _Reshaped) else None
        return _Reshaped(self, self.ex.make_ell_terms_from_dense(self.nm, self.nl, self.Nr, self.R, flat, old=inner))

    # ---- state <-> variables -----------------------------------------------------------------------------------------
    def sync_state_to_device(self):
        for v, m in zip(self.variables, self.vmap):
            if isinstance(v, ConstField):
                if not v._host_dirty:                      # untouched since the last step: the state vector holds it
                    continue
                sc, off, nr = m[0]
                col = np.zeros((1, 2 * self.nm, self.nl, 1))
                if self.m0 == 0:                           # the (m = 0, ell = 0) slot lives on the first rank
                    col[0, 0, 0, 0] = float(v.value.reshape(-1)[0])
                self.ex.assign(self.X4[sc:sc + 1, :, :, off:off + 1], self.ex.from_host(col))
                v._host_dirty = False
                continue
            c = v.require_coeff_space()
            for comp, (sc, off, nr) in enumerate(m):
                self.ex.assign(self.X4[sc:sc + 1, :, :, off:off + nr], c[comp:comp + 1])

    def mark_state_current(self):
        for v, m in zip(self.variables, self.vmap):
            if isinstance(v, ConstField):
                sc, off, nr = m[0]

                src = self.X4[sc:sc + 1, 0:1, 0:1, off:off + 1]
                if self.dist.size > 1:
                    # every rank gets the value INSIDE the step (all ranks are here together): a later read by one rank
                    # alone -- `if rank == 0: print(tau_p['g'])` -- then needs no collective, as in the reference, where a
                    # constant field's data is replicated
                    src = self.dist.pcomm.bcast_scalar(src, self.ex, src_rank=0)

                def pull(src=src):                         # (on demand: a download per step would serialize host and device)
                    return float(np.asarray(self.ex.download(src)).reshape(-1)[0])
                v._pull = pull
                v._host_dirty = False
                continue
            # one coefficient buffer per variable for the life of the solver: a step graph (core/ivp_common.py) replays the
            # copies into the very tensors the fields hold, whichever rotation phase of a multistep scheme recorded them
            bufs = self.__dict__.setdefault("_coeff_bufs", {})
            c = bufs.get(id(v))
            if c is None:
                c = bufs[id(v)] = self.ex.empty(v._cshape())
            for comp, (sc, off, nr) in enumerate(m):
                self.ex.assign(c[comp:comp + 1], self.X4[sc:sc + 1, :, :, off:off + nr])
            v._set_device_coeff(c)

    def evaluate_F(self, out, persistent=False):
        ex = self.ex
        ex.fill_zero(out)
        out4 = out.reshape(self.R, 2 * self.nm, self.nl, self.Nr)
        for eq, m in zip(self.problem.equations, self.emap):
            F = eq["F"]
            if F is None:
                continue
            if isinstance(F, float):
                # constant right-hand side of a scalar equation (e.g. "T(r=Ri) = 1"): the ell = 0 mode, sqrt(2) amplitude
                if eq["rank"] != 0:
                    raise NotImplementedError("constant right-hand side of a tensor equation")
                sc, off, nr = m[0]
                cache = self.__dict__.setdefault("_const_F_cols", {})
                key = (id(eq), sc, off, nr)
                if key not in cache:                       # (uploaded once: no host-to-device copy inside a step)
                    col = np.zeros((1, 2 * self.nm, self.nl, nr))
                    if self.m0 == 0:
                        col[0, 0, 0, 0] = F / SphereBasis.constant_mode_value * \
                            (1.0 if isinstance(eq["basis"], SurfaceBasis) else 1.0 / self._radial_constant(eq["basis"]))
                    cache[key] = ex.from_host(col)
                ex.assign(out4[sc:sc + 1, :, :, off:off + nr], cache[key])
            else:
                c = F.eval_c()
                for comp, (sc, off, nr) in enumerate(m):
                    ex.assign(out4[sc:sc + 1, :, :, off:off + nr], c[comp:comp + 1])

    def _radial_constant(self, basis):
        from ..tools import jacobi
        return float(jacobi.polynomials(1, basis.k + basis.alpha[0], basis.k + basis.alpha[1], np.array([0.0]))[0, 0])


class _Reshaped:
    """device term list applied to the timesteppers' (R, nx, ny) buffers"""

    def __init__(self, solver, dev):
        self.solver, self.dev = solver, dev

    def apply(self, x, y):
        s = self.solver
        shape = (s.R, 2 * s.nm, s.nl, s.Nr)
        self.dev.apply(x.reshape(shape), y.reshape(shape))


class ShellBoundaryValueSolver(ShellSolverBase):
    def __init__(self, problem, **kw):
        super().__init__(problem)
        if self.M_tl.terms:
            raise ValueError("LBVP equations cannot contain time derivatives")
        self._inv = None

    def solve(self):
        if self._inv is None:
            self._inv = self._inverse_terms(0.0, 1.0)
        F = self.ex.zeros((self.R, self.nx, self.ny))
        self.evaluate_F(F)
        self._inv.apply(F, self.X)
        self.mark_state_current()


class ShellInitialValueSolver(IVPLifecycle, ShellSolverBase):
    """IMEX timestepping of M.dt(X) + L.X = F in a shell with the shared schemes of core/timesteppers.py and the shared
    life cycle of core/ivp_common.py::IVPLifecycle (stop conditions incl. stop_wall_time, world clocks, evolve)."""

    def __init__(self, problem, timestepper, enforce_real_cadence=100, warmup_iterations=10, **kw):
        t0 = _time.time()
        ShellSolverBase.__init__(self, problem)
        from . import timesteppers as ts
        from .output import OutputEvaluator
        if isinstance(timestepper, str):
            timestepper = ts.schemes[timestepper]
        self.sim_time = self.initial_sim_time = 0.0
        self._init_lifecycle(enforce_real_cadence, warmup_iterations)
        self._lus = []
        self.timestepper = timestepper(self)
        self.setup_time = _time.time() - t0
        self.evaluator = OutputEvaluator(self)       # analysis handlers: evaluated at the start of a step
        self._step_hooks = [self.evaluator.step_hook]
        self.total_modes = int(self.col_valid.sum()) * 2 * self.nm

    @property
    def state(self):
        return self.variables

    def _hermitian_round_trip(self, f):
        """the state makes a round trip through the dealiased grid (core/solvers.py:675-681)"""
        if isinstance(f, ShellField):
            f.require_grid_space(f.basis.dealias)
            f.require_coeff_space()

    # ---- LHS systems: band LU per ell where the matrices allow it, dense inverses for the rest ---------------------------
    _band = None                 # None: undecided; False: dense inverses only; dict: the band path

    def _band_setup(self):
        """Decide once how a M + b L is solved.  core/ellband.py finds the permutation / recombination that makes the
        per-ell matrices narrow bands; the groups it covers are factorized and swept by csrc/ddh_ellband.hip, the others
        (the ell = 0 system with its dense gauge row) keep a dense inverse.  DDH_SHELL_DENSE=1: dense inverses for all."""
        if self._band is None:
            self._band = False
            import os
            if hasattr(self.ex, "make_ell_band") and os.environ.get("DDH_SHELL_DENSE", "0") != "1":
                from .ellband import EllBandPlan
                prow = sorted({sc for m in self.emap for (sc, off, nr) in m if nr != self.Nr})
                pcol = sorted({sc for m in self.vmap for (sc, off, nr) in m if nr != self.Nr})
                ells = [ell for ell in range(self.nl) if ell >= self.m0]
                plan = EllBandPlan(lambda g: self._dense(self.M_tl, g), lambda g: self._dense(self.L_tl, g),
                                   [self.row_valid[:, g, :] for g in range(self.nl)],
                                   [self.col_valid[:, g, :] for g in range(self.nl)], prow, pcol, self.Nr, ells)
                if plan.per and plan.nbc <= 8 and plan.mp <= 16:
                    limit = [2 * min(max(g - self.m0 + 1, 0), self.nm) for g in range(self.nl)]
                    dev = self.ex.make_ell_band(plan, self.R, 2 * self.nm, self.nl, self.Nr, limit)
                    dinv = None
                    if plan.dense_groups:
                        gs = plan.dense_groups
                        dinv = self.ex.make_dense_inverse([self._dense(self.M_tl, g) for g in gs],
                                                          [self._dense(self.L_tl, g) for g in gs],
                                                          [self.row_valid[:, g, :].reshape(-1) for g in gs],
                                                          [self.col_valid[:, g, :].reshape(-1) for g in gs], complex_=False)
                    self._band = dict(plan=plan, dev=dev, dinv=dinv)
                    logger.info("shell LHS: band LU for %d of %d ell (kl %d, ku %d); dense inverse for ell in %s" % (
                        len(plan.per), len(ells), plan.kl, plan.ku, plan.dense_groups))
        return self._band

    def factor(self, a, b, reuse=-1):
        old = self._lus[reuse] if (reuse is not None and reuse >= 0) else None
        band = self._band_setup()
        if band:
            index = band["dev"].factor(a, b, index=old["index"] if isinstance(old, dict) else None)
            dense = None
            if band["dinv"] is not None:
                flat = band["dinv"].compute(a, b)
                dense = old["dense"] if isinstance(old, dict) else self.ex.empty(tuple(flat.shape))
                self.ex.assign(dense, flat)
            inv = dict(index=index, dense=dense)
        else:
            inv = self._inverse_terms(a, b, old=old)
        if not hasattr(self, "_lu_params"):
            self._lu_params = {}
        if reuse is not None and reuse >= 0:
            self._lus[reuse] = inv
            self._lu_params[reuse] = (float(a), float(b))
            return reuse
        self._lus.append(inv)
        self._lu_params[len(self._lus) - 1] = (float(a), float(b))
        return len(self._lus) - 1

    def solve(self, lu, rhs, x):
        inv = self._lus[lu]
        if isinstance(inv, dict):
            band = self._band
            self.ex.fill_zero(x)
            band["dev"].solve(inv["index"], rhs, x)
            if inv["dense"] is not None:
                shape = (self.R, 2 * self.nm, self.nl, self.Nr)
                n2 = (self.R * self.Nr) ** 2
                for k, g in enumerate(band["plan"].dense_groups):
                    self.ex.dense_group_solve(inv["dense"][k * n2:(k + 1) * n2], rhs.reshape(shape), x.reshape(shape), g)
        else:
            inv.apply(rhs, x)
        probe = getattr(self, "solve_probe", None)
        if probe is not None:                        # parity checks: keep (a, b, rhs, x) of every solve
            a, b = self._lu_params[lu]
            probe.append(dict(a=a, b=b, rhs=self.ex.download(rhs).copy(), x=self.ex.download(x).copy()))



# ==================================================================================================
# products with radial non-constant coefficients (NCCs) on the LHS
# ==================================================================================================

def gamma_regularity(rank_a, rank_b, ell_a, ell_b, ell_c, contract=False):
    """Gamma[alpha, beta, gamma]: regularity components of a product, out_gamma = sum Gamma a_alpha b_beta
    (Product.Gamma, core/arithmetic.py:560-580: the coordinate-component product tensor carried through the spin
    intertwiner U and the regularity intertwiners Q(ell) of the three operands).  contract: DotProduct of the last
    index of a with the first of b (GammaCoord :649-664), else the tensor product (MultiplyFields)."""
    ia_list = reg_indices(rank_a)
    ib_list = reg_indices(rank_b)
    rank_c = rank_a + rank_b - (2 if contract else 0)
    ic_list = reg_indices(rank_c)
    G = np.zeros((len(ia_list), len(ib_list), len(ic_list)), dtype=complex)
    for ia, a in enumerate(ia_list):
        for ib, b in enumerate(ib_list):
            if contract:
                if a[-1] != b[0]:
                    continue
                c = tuple(a[:-1]) + tuple(b[1:])
            else:
                c = tuple(a) + tuple(b)
            G[ia, ib, ic_list.index(c)] = 1.0
    # coordinate -> spin components
    UA = np.conj(SphericalCoordinates.U_forward(rank_a))
    UB = np.conj(SphericalCoordinates.U_forward(rank_b))
    UC = SphericalCoordinates.U_forward(rank_c)
    G = np.einsum("ai,bj,ck,ijk->abc", UA, UB, UC, G)
    # spin -> regularity components
    QA = sph.intertwiner(ell_a, rank_a).T if rank_a else np.eye(1)
    QB = sph.intertwiner(ell_b, rank_b).T if rank_b else np.eye(1)
    QC = sph.intertwiner(ell_c, rank_c).T if rank_c else np.eye(1)
    return np.einsum("ai,bj,ck,ijk->abc", QA, QB, QC, G)
