"""
The sphere: S2 coordinates, SphereBasis, tensor fields, the linear operators of the shallow-water family and
the per-m IMEX / boundary-value solvers (BASELINE config 4).

What the reference does (file:line under /root/reference/dedalus) and what happens here:

* Layout.  The reference stores real-dtype sphere coefficients in a triangular-truncation *packed* array
  (SphereBasis.elements_to_groups, core/basis.py:2839-2891).  On the device a field is simply
  [component][2 m + part][ell] (part = cos / msin of azimuthal mode m; spin components in coefficient
  space, coordinate components on the grid); `field['c']` converts to and from the reference's packed
  layout on the host, so user code and the parity fixtures see the reference's arrays.
* Transforms (SphereBasis.forward/backward_transform_*, core/basis.py:3068-3153): azimuth = the RealFourier
  FFT kernel (strided), spin recombination (core/basis.py:1595-1663) = ddh_spin_recombine, colatitude =
  the grouped SWSH GEMV kernel (row a12, csrc/ddh_swsh.hip), one launch per spin weight.
* Linear operators.  SphereGradient / SphereDivergence / SphereLaplacian / SphereAverage /
  ConvertConstantSphere symbols (core/basis.py:3279-3420, 5296-5320, SphereBasis.k :3156-3158), SpinSkew
  (core/operators.py:2125-2147) and MulCosine (:2995-3046; Jacobi 'Z' of libraries/dedalus_sphere/sphere.py:88-92)
  are ell-local, so an operator -- and any composition of them -- is a list of terms
        out[co][m][ell] += coef[m][ell] * in[ci][m][ell + d]
  (class TermList), applied on the device by ddh_sphere_terms_apply.  The per-m subproblem matrices of
  the reference (core/subsystems.py:497-596) are the same terms laid out per m.
* Solvers.  Per m, all variables x (ell >= m) form one small complex system; invalid modes
  (SphereBasis.valid_elements, core/basis.py:3183-3211) are dropped exactly like the reference's
  valid-mode filtering.  LHS = a M + b L is inverted per m on the host when a, b change and applied with
  ddh_cgemv_batch_apply; the IMEX schemes are the shared ones of core/timesteppers.py.
"""

import logging
import numbers
import time as _time

import numpy as np

from ..tools import jacobi
from ..tools import sphere as sph
from .coords import Coordinate
from .ivp_common import IVPLifecycle


logger = logging.getLogger(__name__)

# ==================================================================================================
# coordinates, basis, distributor
# ==================================================================================================

class S2Coordinates:
    """(azimuth, colatitude); spin component ordering (-, +)  (core/coords.py:201-252)."""
    dim = 2
    spin_ordering = (-1, +1)

    def __init__(self, azimuth, colatitude):
        self.names = (azimuth, colatitude)
        self.azimuth = Coordinate(azimuth, cs=self)
        self.colatitude = Coordinate(colatitude, cs=self)
        self.coords = (self.azimuth, self.colatitude)

    def __iter__(self):
        return iter(self.coords)

    @staticmethod
    def U_forward(order=1):
        """Unitary map from coordinate (phi, theta) to spin (-, +) components: u[+-] = (u[theta] +- i u[phi]) / sqrt 2."""
        U = np.array([[-1j, 1], [+1j, 1]]) / np.sqrt(2)
        out = np.array([[1.0 + 0j]])
        for _ in range(order):
            out = np.kron(out, U)
        return out


class SphereBasis:
    """Spin-weighted spherical harmonics on S2, real dtype (core/basis.py:2672-2771)."""

    def __init__(self, coordsys, shape, dtype=np.float64, radius=1, dealias=(1, 1), azimuth_library=None,
                 colatitude_library=None):
        if not isinstance(coordsys, S2Coordinates):
            raise ValueError("Sphere coordsys must be S2Coordinates.")
        if np.dtype(dtype) != np.float64:
            raise NotImplementedError("sphere fields: float64 only")
        shape = tuple(int(s) for s in shape)
        if len(shape) != 2:
            raise ValueError("Sphere shape must have length 2.")
        if radius < 0:
            raise ValueError("Sphere radius must be non-negative.")
        if isinstance(dealias, numbers.Number):
            dealias = (dealias,) * 2
        if shape[0] % 4 != 0:
            raise ValueError("Don't use a phi resolution that isn't divisible by 4, please")
        self.coordsys, self.shape, self.dtype = coordsys, shape, np.dtype(dtype)
        self.radius, self.dealias = radius, tuple(float(d) for d in dealias)
        self.Nphi, self.Ntheta = shape
        self.Lmax = max(0, self.Ntheta - 2)
        self.mmax = self.Nphi // 2 - 1
        self.nm = self.Nphi // 2                 # azimuthal wavenumbers 0 .. Nphi/2 - 1 (cos/msin pairs)
        self.nl = self.Lmax + 1
        self.volume = 4 * np.pi * radius ** 2
        self._plans = {}

    def m_range(self):
        """(first local m, number of local m): block distribution of the azimuthal wavenumbers over the ranks of the
        distributor's 1-D mesh (the reference distributes the first coefficient axis, core/distributor.py:357-385)."""
        d = getattr(self.coordsys, "dist", None)
        P, r = (int(getattr(d, "size", 1)), int(getattr(d, "rank", 0))) if d is not None else (1, 0)
        if self.nm % P:
            raise ValueError("%d azimuthal wavenumbers do not divide over %d ranks" % (self.nm, P))
        return r * (self.nm // P), self.nm // P

    @property
    def m0(self):
        return self.m_range()[0]

    @property
    def nml(self):
        return self.m_range()[1]

    constant_mode_value = 1 / np.sqrt(2)

    def grid_shape(self, scales):
        return (int(np.ceil(scales[0] * self.Nphi)), int(np.ceil(scales[1] * self.Ntheta)))

    def grids(self, scales):
        Np, Nt = self.grid_shape(scales)
        phi = 2 * np.pi * np.arange(Np) / Np
        z, _ = sph.quadrature(Nt)
        theta = np.arccos(np.asarray(z, dtype=np.float64))
        return phi, theta

    # ---- the reference's packed coefficient layout ------------------------------------------------------------
    def packed_shape(self):
        return (self.Nphi // 2, self.Lmax + 1 + max(0, self.Lmax + 2 - self.Nphi // 2))

    def packed_groups(self):
        """(m, ell) of every element (i, j) of the packed coefficient array, restating
        SphereBasis.elements_to_groups for float64 (core/basis.py:2868-2889)."""
        i, j = np.indices(self.packed_shape())
        Nphi, Lmax = self.Nphi, self.Lmax
        shift = max(0, Lmax + 2 - Nphi // 2)
        m = i // 2
        ell = j - shift
        neg = ell < m
        m = np.where(neg, (Nphi // 2 - 1) - m, m)
        ell = np.where(neg, Lmax - j, ell)
        mz = i < 2
        m = np.where(mz, 0, m)
        ell = np.where(mz, j, ell)
        mm = (i < 2) & (j > Lmax)
        m = np.where(mm, Nphi // 2 - 1, m)
        ell = np.where(mm, j - shift, ell)
        return m, ell

    def pack_index(self):
        """Index arrays (rows, cols, valid): packed[i, j] = natural[2 m + (i % 2), ell] where 0 <= ell <= Lmax."""
        if "pack" not in self._plans:
            m, ell = self.packed_groups()
            i, _ = np.indices(self.packed_shape())
            ok = (ell >= 0) & (ell <= self.Lmax) & (m >= 0) & (m < self.nm)
            rows = np.where(ok, 2 * m + (i % 2), 0)
            cols = np.where(ok, ell, 0)
            self._plans["pack"] = (rows, cols, ok)
        return self._plans["pack"]

    def packed_ell_rows(self):
        """(ell, i0, i1, j, j+1) rows of the reference's ell_maps for the packed layout (core/basis.py:2983-3011):
        per ell and packed column the BOUNDING BOX of the rows holding that ell (local_groupset_slices,
        core/distributor.py:460-494).  Boxes of different ell overlap in the folded part of the layout; the
        reference recombines a slot once per covering box, which the shell's regularity tables reproduce."""
        m, ell = self.packed_groups()
        rows = []
        for lg in range(self.Lmax + 1):
            for j in range(m.shape[1]):
                hit = np.where(ell[:, j] == lg)[0]
                if hit.size:
                    rows.append((lg, int(hit.min()), int(hit.max()) + 1, j, j + 1))
        return np.array(rows, dtype=np.int64).reshape(-1, 5)

    # ---- spin bookkeeping ---------------------------------------------------------------------------------------
    @staticmethod
    def spin_indices(rank):
        return list(np.ndindex(*((2,) * rank)))

    @staticmethod
    def spin_totals(rank):
        return [sum((-1, +1)[a] for a in idx) for idx in np.ndindex(*((2,) * rank))]

    @staticmethod
    def k(ell, s, mu):
        """core/basis.py:3156-3158"""
        return -mu * np.sqrt(np.maximum(0, (ell - mu * s) * (ell + mu * s + 1) / 2))

    def valid(self, rank):
        """[ncomp][nm][nl] validity of coefficient modes (SphereBasis.valid_elements, core/basis.py:3183-3211):
        ell >= max(|m|, |s|); the msin part of ell == 0 is dropped for scalars and vectors (handled by the
        solvers, which keep z(m=0, ell=0) real)."""
        m = np.arange(self.nm)[:, None]
        ell = np.arange(self.nl)[None, :]
        out = []
        for s in self.spin_totals(rank):
            out.append(ell >= np.maximum(m, abs(s)))
        return np.array(out)

    def cos_bands(self, s):
        """MulCosine for spin weight s as three coefficient arrays {d: coef[m][ell]} (d = -1, 0, +1):
        out[ell] = sum_d coef_d[m, ell] in[ell + d]; the Jacobi 'Z' matrix of the (|m+s|, |m-s|) family."""
        key = ("cos", s)
        if key not in self._plans:
            bands = {d: np.zeros((self.nm, self.nl)) for d in (-1, 0, 1)}
            for m in range(self.nm):
                Lmin = max(m, abs(s))
                n = self.Lmax + 1 - Lmin
                if n < 1:
                    continue
                J = jacobi.jacobi_matrix(n, abs(m + s), abs(m - s))
                J = np.asarray(J.toarray() if hasattr(J, "toarray") else J, dtype=np.float64)
                bands[0][m, Lmin:] = np.diag(J)
                if n > 1:
                    bands[+1][m, Lmin:self.Lmax] = np.diag(J, 1)
                    bands[-1][m, Lmin + 1:] = np.diag(J, -1)
            self._plans[key] = bands
        return self._plans[key]

    # ---- device transform plans -----------------------------------------------------------------------------------
    def colatitude_plan(self, ex, Ntheta_g, rank):
        """One grouped-GEMV plan for ALL spin components of a rank-`rank` tensor: the component axis is folded into
        the m axis of the reduced views ([1][ncomp * 2 nm][n][1]), every (component, m) is one group with the
        matrix pair of (m, spin weight of the component).  The reference transforms component by component
        (core/basis.py:3119-3122, 3145-3148); one launch per field streams all matrices at once."""
        key = ("swsh", id(ex), Ntheta_g, rank)
        if key not in self._plans:
            import os
            spins = self.spin_totals(rank)
            # Components of opposite spin weight share their matrices: F_{-s}[l, j] = (-1)^(l + m) F_{+s}[l, N-1-j]
            # (and two components of equal spin weight trivially do).  On the device the partner rides along as a
            # second right-hand-side set of the +s group (ddh_grouped_mmt_set_pairs): the transform is bound by
            # streaming the matrices, which are then stored and read once instead of twice.
            pairs = []                                     # (primary component, partner or None, mode)
            if getattr(ex, "mmt_pairs", False) and os.environ.get("DDH_SWSH_PAIRS", "1") != "0":
                by_spin = {}
                for i, sv in enumerate(spins):
                    by_spin.setdefault(sv, []).append(i)
                used = set()
                for sv in sorted(by_spin, reverse=True):
                    if sv > 0:
                        for a, b in zip(by_spin[sv], by_spin.get(-sv, [])):
                            pairs.append((a, b, 2))
                            used.update((a, b))
                    elif sv == 0:
                        z = by_spin[0]
                        for a, b in zip(z[0::2], z[1::2]):
                            pairs.append((a, b, 1))
                            used.update((a, b))
                pairs += [(i, None, 0) for i in range(len(spins)) if i not in used]
            else:
                pairs = [(i, None, 0) for i in range(len(spins))]
            groups, keys, fwd, bwd = [], [], [], []
            pg, pc, pm, pp = [], [], [], []
            cache = {}
            for (i, partner, mode) in pairs:
                sv = spins[i]
                for m in range(self.nm):
                    mk = m + 4096 * (sv + 8)                   # matrix key of (m, s)
                    ne = max(self.Lmax + 1 - m, 0)
                    groups.append((mk if ne > 0 else -1 - m, i * 2 * self.nm + 2 * m, i * 2 * self.nm + 2 * m, 2, m, 1, ne))
                    j = partner if partner is not None else 0
                    pg.append(j * 2 * self.nm + 2 * m)
                    pc.append(j * 2 * self.nm + 2 * m)
                    pm.append(mode)
                    pp.append(m & 1)
                    if ne > 0 and mk not in cache:
                        cache[mk] = sph.swsh_matrices(Ntheta_g, self.Lmax, m, sv)
                        keys.append(mk)
                        fwd.append(cache[mk][0])
                        bwd.append(cache[mk][1])
            plan = ex.make_grouped_mmt(Ntheta_g, np.array(groups, dtype=np.int64), keys, fwd, bwd)
            if any(pm):
                plan.set_pairs(pg, pc, pm, pp)
            self._plans[key] = plan
        return self._plans[key]

    def recombination_matrix(self, rank, forward):
        """Real (2 ncomp x 2 ncomp) matrix on (component, cos/msin) pairs (spin_recombination_matrix,
        core/basis.py:1576-1593): coordinate -> spin components (forward) or back."""
        U = S2Coordinates.U_forward(rank)
        if not forward:
            U = U.T.conj()
        return np.kron(U.real, np.eye(2)) + np.kron(U.imag, np.array([[0.0, -1.0], [1.0, 0.0]]))

    def __eq__(self, other):
        return isinstance(other, SphereBasis) and (self.shape, self.radius, self.dealias) == (
            other.shape, other.radius, other.dealias)

    def __hash__(self):
        return hash((self.shape, self.radius, self.dealias))


class SphereDistributor:
    """Distributor for an S2 coordinate system (single device: the m-sharded mesh of the reference,
    core/distributor.py:60-70, is not built for the sphere in this round)."""

    def __init__(self, coordsys, comm=None, mesh=None, dtype=None, executor=None):
        self.coordsystems = (coordsys,)
        self.coordsys = coordsys
        self.coords = coordsys.coords
        self.dim = 2
        self.dtype = np.dtype(np.float64 if dtype is None else dtype)
        if self.dtype != np.float64:
            raise NotImplementedError("sphere fields: float64 only")
        if mesh is not None and int(np.prod(mesh)) > 1:
            raise NotImplementedError("sphere problems run on one device in this round")
        self.mesh, self.size, self.rank, self.comm = (), 1, 0, comm
        self._executor = executor

    @property
    def executor(self):
        if self._executor is None:
            from ..executor import HipExecutor
            self._executor = HipExecutor()       # raises without a gfx950 device: no CPU fallback
        return self._executor

    def Field(self, name=None, bases=None, tensorsig=None, dtype=None):
        rank = len(tensorsig) if tensorsig else 0
        if isinstance(bases, (tuple, list)):
            bases = bases[0] if bases else None
        return SField(self, basis=bases, rank=rank, name=name)

    ScalarField = Field

    def VectorField(self, coordsys, name=None, bases=None, dtype=None):
        return self.Field(name=name, bases=bases, tensorsig=(coordsys,))

    def TensorField(self, coordsys, name=None, bases=None, order=2, dtype=None):
        sig = tuple(coordsys) if isinstance(coordsys, (tuple, list)) else (coordsys,) * order
        return self.Field(name=name, bases=bases, tensorsig=sig)

    def local_grids(self, *bases, scales=None):
        basis = bases[0]
        if scales is None:
            scales = (1, 1)
        elif isinstance(scales, numbers.Number):
            scales = (scales, scales)
        phi, theta = basis.grids(scales)
        return phi[:, None], theta[None, :]

    def local_grid(self, basis, scale=None):
        raise NotImplementedError("use local_grids for the sphere")


# ==================================================================================================
# term lists
# ==================================================================================================

class TermList:
    """Linear map between coefficient arrays: out[co][m][ell] = sum coef[m][ell] * in[ci][m][ell + d]."""

    def __init__(self, ncomp_out, ncomp_in, terms=None):
        self.ncomp_out, self.ncomp_in = ncomp_out, ncomp_in
        self.terms = list(terms or [])          # (co, ci, d, coef complex [nm][nl])

    @staticmethod
    def identity(ncomp, nm, nl):
        return TermList(ncomp, ncomp, [(c, c, 0, np.ones((nm, nl), dtype=complex)) for c in range(ncomp)])

    def scaled(self, a):
        return TermList(self.ncomp_out, self.ncomp_in, [(co, ci, d, a * cf) for (co, ci, d, cf) in self.terms])

    def __add__(self, other):
        assert (self.ncomp_out, self.ncomp_in) == (other.ncomp_out, other.ncomp_in)
        return TermList(self.ncomp_out, self.ncomp_in, self.terms + other.terms).merged()

    def merged(self):
        acc = {}
        for (co, ci, d, cf) in self.terms:
            key = (co, ci, d)
            acc[key] = acc[key] + cf if key in acc else np.array(cf, dtype=complex)
        return TermList(self.ncomp_out, self.ncomp_in,
                        [(co, ci, d, cf) for (co, ci, d), cf in sorted(acc.items()) if np.any(cf != 0)])

    def compose(self, inner):
        """self o inner"""
        assert self.ncomp_in == inner.ncomp_out
        out = []
        for (co, cm, d1, c1) in self.terms:
            for (cm2, ci, d2, c2) in inner.terms:
                if cm2 != cm:
                    continue
                nl = c1.shape[1]
                sh = np.zeros_like(c2)
                if d1 >= 0:
                    sh[:, :nl - d1] = c2[:, d1:]
                else:
                    sh[:, -d1:] = c2[:, :nl + d1]
                out.append((co, ci, d1 + d2, c1 * sh))
        return TermList(self.ncomp_out, inner.ncomp_in, out).merged()

    def embed(self, row0, col0, nrows, ncols):
        return TermList(nrows, ncols, [(co + row0, ci + col0, d, cf) for (co, ci, d, cf) in self.terms])


def _mask(basis, s_list):
    m = np.arange(basis.nm)[:, None]
    ell = np.arange(basis.nl)[None, :]
    return [ell >= np.maximum(m, abs(s)) for s in s_list]


def op_termlist(kind, basis, rank_in, **kw):
    """Term list of one elementary operator acting on a rank-`rank_in` tensor (spin components)."""
    nm, nl, R = basis.nm, basis.nl, basis.radius
    ell = np.arange(nl)[None, :] + np.zeros((nm, 1))
    idx_in = basis.spin_indices(rank_in)
    s_in = basis.spin_totals(rank_in)
    vin = _mask(basis, s_in)
    if kind == "grad":
        idx_out = basis.spin_indices(rank_in + 1)
        terms = []
        for ci, (tau, s) in enumerate(zip(idx_in, s_in)):
            for a, mu in enumerate((-1, +1)):
                co = idx_out.index((a,) + tuple(tau))
                kk = basis.k(ell, s, mu) / R
                kk = kk * (np.abs(s) <= ell) * (np.abs(s + mu) <= ell) * vin[ci] * (ell >= np.arange(nm)[:, None])
                terms.append((co, ci, 0, kk.astype(complex)))
        return TermList(len(idx_out), len(idx_in), terms)
    if kind == "div":
        idx_out = basis.spin_indices(rank_in - 1)
        s_out = basis.spin_totals(rank_in - 1)
        terms = []
        for ci, (tau, s) in enumerate(zip(idx_in, s_in)):
            co = idx_out.index(tuple(tau[1:]))
            mu = s_out[co] - s
            kk = basis.k(ell, s, mu) / R
            kk = kk * (np.abs(s) <= ell) * (np.abs(s_out[co]) <= ell) * vin[ci]
            terms.append((co, ci, 0, kk.astype(complex)))
        return TermList(len(idx_out), len(idx_in), terms)
    if kind == "lap":
        terms = []
        for ci, s in enumerate(s_in):
            k = basis.k
            kl = (k(ell, s + 1, -1) * k(ell, s, +1) + k(ell, s - 1, +1) * k(ell, s, -1)) / R ** 2
            kl = kl * (np.abs(s) <= ell) * vin[ci]
            terms.append((ci, ci, 0, kl.astype(complex)))
        return TermList(len(idx_in), len(idx_in), terms)
    if kind == "skew":
        if rank_in < 1:
            raise ValueError("skew needs a vector")
        terms = []
        for ci, tau in enumerate(idx_in):
            sign = -1.0 if tau[0] == 0 else +1.0
            terms.append((ci, ci, 0, (1j * sign) * vin[ci].astype(complex)))
        return TermList(len(idx_in), len(idx_in), terms)
    if kind == "mulcos":
        terms = []
        for ci, s in enumerate(s_in):
            bands = basis.cos_bands(s)
            for d in (-1, 0, 1):
                terms.append((ci, ci, d, bands[d].astype(complex)))
        return TermList(len(idx_in), len(idx_in), terms)
    if kind == "ave":          # SphereAverage: keeps the ell = 0 coefficient (of m = 0)
        terms = []
        for ci in range(len(idx_in)):
            cf = np.zeros((nm, nl), dtype=complex)
            if s_in[ci] == 0:
                cf[0, 0] = 1.0
            terms.append((ci, ci, 0, cf))
        return TermList(len(idx_in), len(idx_in), terms)
    if kind == "convert_constant":     # ConvertConstantSphere: constant -> ell = 0 mode with unit amplitude sqrt 2
        cf = np.zeros((nm, nl), dtype=complex)
        cf[0, 0] = 1.0 / SphereBasis.constant_mode_value
        return TermList(1, 1, [(0, 0, 0, cf)])
    raise ValueError(kind)


# ==================================================================================================
# operands
# ==================================================================================================

class NonlinearError(ValueError):
    pass


class SOperand:
    """Expression node on the sphere.  rank = tensor rank (components over S2), basis may be None (constant)."""
    __array_priority__ = 100.0
    __array_ufunc__ = None          # numpy scalars defer to __rmul__ / __radd__

    def __add__(self, other):
        return SAdd.make(self, other)

    __radd__ = __add__

    def __sub__(self, other):
        return SAdd.make(self, -1 * other if isinstance(other, SOperand) else -other)

    def __rsub__(self, other):
        return SAdd.make(-1 * self, other)

    def __neg__(self):
        return SScale(-1.0, self)

    def __pos__(self):
        return self

    def __mul__(self, other):
        if isinstance(other, numbers.Number):
            return SScale(other, self)
        if isinstance(other, SOperand):
            return SProduct(self, other)
        return NotImplemented

    def __rmul__(self, other):
        if isinstance(other, numbers.Number):
            return SScale(other, self)
        return NotImplemented

    def __truediv__(self, other):
        if isinstance(other, numbers.Number):
            return SScale(1.0 / other, self)
        return NotImplemented

    def __matmul__(self, other):
        return SDot(self, other)

    @property
    def ncomp(self):
        return 2 ** self.rank

    def evaluate(self):
        f = SField(self.dist, basis=self.basis, rank=self.rank)
        f._set_device_coeff(self.eval_c())
        return f

    def has_dt(self):
        return any(a.has_dt() for a in getattr(self, "args", ()) if isinstance(a, SOperand))

    # evaluation to device arrays ------------------------------------------------------------------------------
    def eval_c(self):
        """Device coefficient array [ncomp][2 nm][nl] (a fresh array or a field's own, never to be modified)."""
        g = self.eval_g()
        return _forward(self.dist, self.basis, self.rank, g, self.basis.dealias)

    def eval_g(self):
        """Device grid array [ncomp][Nphi_g][Ntheta_g] at the dealias scales."""
        return _backward(self.dist, self.basis, self.rank, self.eval_c(), self.basis.dealias)


def _ex(dist):
    return dist.executor


def _backward(dist, basis, rank, c, scales):
    ex = _ex(dist)
    nc = 2 ** rank
    Np, Nt = basis.grid_shape(scales)
    t1 = ex.empty((nc, 2 * basis.nm, Nt))
    basis.colatitude_plan(ex, Nt, rank).backward(c.reshape(1, nc * 2 * basis.nm, basis.nl, 1),
                                                 t1.reshape(1, nc * 2 * basis.nm, Nt, 1))
    if rank > 0:
        t2 = ex.empty((nc, 2 * basis.nm, Nt))
        ex.spin_recombine(t1, t2, basis.recombination_matrix(rank, forward=False))
    else:
        t2 = t1
    g = ex.empty((nc, Np, Nt))
    ex.transform(("rfft", Np, basis.Nphi), None, "backward", t2, g, nc, Nt)
    return g


def _forward(dist, basis, rank, g, scales):
    ex = _ex(dist)
    nc = 2 ** rank
    Np, Nt = basis.grid_shape(scales)
    t1 = ex.empty((nc, 2 * basis.nm, Nt))
    ex.transform(("rfft", Np, basis.Nphi), None, "forward", g, t1, nc, Nt)
    if rank > 0:
        t2 = ex.empty((nc, 2 * basis.nm, Nt))
        ex.spin_recombine(t1, t2, basis.recombination_matrix(rank, forward=True))
    else:
        t2 = t1
    c = ex.zeros((nc, 2 * basis.nm, basis.nl))
    basis.colatitude_plan(ex, Nt, rank).forward(t2.reshape(1, nc * 2 * basis.nm, Nt, 1),
                                                c.reshape(1, nc * 2 * basis.nm, basis.nl, 1))
    return c


class SField(SOperand):
    def __init__(self, dist, basis=None, rank=0, name=None):
        self.dist, self.basis, self.rank, self.name = dist, basis, rank, name
        self.scales = (1.0, 1.0)
        self._c = None              # device coefficients [ncomp][2 nm][nl] (basis) or host scalar (constant)
        self._g = None
        self._g_scales = None
        self.layout = "c"
        self._host = None
        self._host_layout = None
        self._host_scales = None
        self._authority = "device"
        self._const = np.zeros(())   # constants (no basis): their value
        self.args = ()

    def __repr__(self):
        return self.name or "<SField %d>" % id(self)

    @property
    def tensorsig(self):
        return (self.dist.coordsys,) * self.rank

    @property
    def ex(self):
        return self.dist.executor

    # ---- storage -------------------------------------------------------------------------------------------------
    def _cshape(self):
        return (self.ncomp, 2 * self.basis.nm, self.basis.nl)

    def _alloc_c(self):
        if self._c is None:
            self._c = self.ex.zeros(self._cshape())
        return self._c

    def _set_device_coeff(self, c):
        self._c = c
        self.layout = "c"
        self._authority = "device"

    def _user_shape(self, layout, scales):
        t = (2,) * self.rank
        if self.basis is None:
            return t + (1, 1)
        if layout == "g":
            return t + self.basis.grid_shape(scales)
        return t + self.basis.packed_shape()

    def _remedy(self, scales):
        if scales is None:
            return (1.0, 1.0)
        if isinstance(scales, numbers.Number):
            return (float(scales),) * 2
        return tuple(float(s) for s in scales)

    def _sync_to_device(self):
        if self._authority != "host":
            return
        self._authority = "device"
        lay, sc = self._host_layout, self._host_scales
        if self.basis is None:
            self._const = np.array(self._host).reshape(()) * (1.0 if lay == "g" else SphereBasis.constant_mode_value)
            return
        if lay == "c":
            rows, cols, ok = self.basis.pack_index()
            nat = np.zeros(self._cshape())
            for c in range(self.ncomp):
                nat[c][rows[ok], cols[ok]] = self._host.reshape((self.ncomp,) + rows.shape)[c][ok]
            self.ex.upload(self._alloc_c(), nat)
            self.layout = "c"
        else:
            Np, Nt = self.basis.grid_shape(sc)
            if self._g is None or self._g_scales != sc:
                self._g = self.ex.empty((self.ncomp, Np, Nt))
                self._g_scales = sc
            self.ex.upload(self._g, np.ascontiguousarray(self._host.reshape(self.ncomp, Np, Nt)))
            self.layout = "g"
            self.scales = sc

    def require_coeff_space(self):
        self._sync_to_device()
        if self.basis is None:
            return None
        if self.layout == "g":
            self._c = _forward(self.dist, self.basis, self.rank, self._g, self._g_scales)
            self.layout = "c"
        return self._alloc_c()

    def require_grid_space(self, scales=None):
        self._sync_to_device()
        scales = self._remedy(scales)
        if self.layout == "g" and self._g_scales == scales:
            return self._g
        c = self.require_coeff_space()
        self._g = _backward(self.dist, self.basis, self.rank, c, scales)
        self._g_scales = scales
        self.layout = "g"
        self.scales = scales
        return self._g

    def change_scales(self, scales):
        scales = self._remedy(scales)
        if scales == self.scales:
            return
        self._sync_to_device()
        if self.basis is not None and self.layout == "g":
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
            if self.basis is None:
                v = float(self._const) * (1.0 if layout == "g" else 1.0 / SphereBasis.constant_mode_value)
                self._host = np.full(shape, v)
            elif layout == "c":
                nat = np.asarray(self.ex.download(self.require_coeff_space()))
                rows, cols, ok = self.basis.pack_index()
                out = np.zeros((self.ncomp,) + rows.shape)
                for c in range(self.ncomp):
                    out[c][ok] = nat[c][rows[ok], cols[ok]]
                self._host = out.reshape(shape)
            else:
                g = self.require_grid_space(self.scales)
                self._host = np.array(self.ex.download(g)).reshape(shape)
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

    @property
    def data(self):
        return self[self.layout if self._authority == "device" else self._host_layout]

    # ---- evaluation / linearisation ------------------------------------------------------------------------------
    def eval_c(self):
        if self.basis is None:
            raise ValueError("constant field has no coefficient array")
        return self.require_coeff_space()

    def eval_g(self):
        if self.layout == "g" and self._authority == "device" and self._g_scales == self.basis.dealias:
            return self._g
        self._sync_to_device()
        cache = getattr(self.dist, "_grid_cache", None)       # one backward transform per field and RHS pass
        if cache is not None and id(self) in cache:
            return cache[id(self)]
        g = _backward(self.dist, self.basis, self.rank, self.require_coeff_space(), self.basis.dealias)
        if cache is not None:
            cache[id(self)] = g
        return g

    def lin(self, variables, basis):
        """{variable index: TermList acting on that variable's coefficients}; raises if not a problem variable."""
        for i, v in enumerate(variables):
            if v is self:
                if self.basis is None:
                    return {i: op_termlist("convert_constant", basis, 0)}, False
                return {i: TermList.identity(self.ncomp, basis.nm, basis.nl)}, False
        raise NonlinearError("%r is not a problem variable" % (self,))

    def has_dt(self):
        return False

    def evaluate(self):
        return self


def _basis_of(*ops):
    for o in ops:
        if isinstance(o, SOperand) and o.basis is not None:
            return o.basis
    return None


class SScale(SOperand):
    def __init__(self, a, arg):
        self.a, self.arg, self.args = float(a), arg, (arg,)
        self.dist, self.basis, self.rank = arg.dist, arg.basis, arg.rank

    def eval_c(self):
        ex = _ex(self.dist)
        c = self.arg.eval_c()
        out = ex.empty(tuple(c.shape))
        ex.lincomb(out, [c], [self.a])
        return out

    def eval_g(self):
        ex = _ex(self.dist)
        g = self.arg.eval_g()
        out = ex.empty(tuple(g.shape))
        ex.lincomb(out, [g], [self.a])
        return out

    def lin(self, variables, basis):
        d, dt = self.arg.lin(variables, basis)
        return {i: tl.scaled(self.a) for i, tl in d.items()}, dt


class SAdd(SOperand):
    @staticmethod
    def make(a, b):
        if isinstance(b, numbers.Number):
            if b == 0:
                return a
            raise NotImplementedError("adding a number to a sphere field")
        if isinstance(a, numbers.Number):
            if a == 0:
                return b
            raise NotImplementedError("adding a number to a sphere field")
        return SAdd(a, b)

    def __init__(self, a, b):
        if a.rank != b.rank:
            raise ValueError("cannot add tensors of different rank")
        self.args = (a, b)
        self.dist, self.rank = a.dist, a.rank
        self.basis = _basis_of(a, b)

    def _coef_of(self, a):
        """coefficient array of an argument, converting constants (no basis) to the ell = 0 mode"""
        if a.basis is None:
            f = a.evaluate() if not isinstance(a, SField) else a
            f._sync_to_device()
            nat = np.zeros((1, 2 * self.basis.nm, self.basis.nl))
            nat[0, 0, 0] = float(f._const) / SphereBasis.constant_mode_value
            return _ex(self.dist).from_host(nat)
        return a.eval_c()

    def eval_c(self):
        ex = _ex(self.dist)
        cs = [self._coef_of(a) for a in self.args]
        out = ex.empty(tuple(cs[0].shape))
        ex.lincomb(out, cs, [1.0, 1.0])
        return out

    def eval_g(self):
        if any(a.basis is None for a in self.args):
            return SOperand.eval_g(self)
        ex = _ex(self.dist)
        gs = [a.eval_g() for a in self.args]
        out = ex.empty(tuple(gs[0].shape))
        ex.lincomb(out, gs, [1.0, 1.0])
        return out

    def lin(self, variables, basis):
        out, anydt = {}, None
        for a in self.args:
            d, dt = a.lin(variables, basis)
            if anydt is None:
                anydt = dt
            elif anydt != dt:
                raise NonlinearError("cannot mix dt and non-dt terms inside one sum node; write them as separate terms")
            for i, tl in d.items():
                out[i] = out[i] + tl if i in out else tl
        return out, bool(anydt)


class SLinear(SOperand):
    """grad / div / lap / skew / MulCosine / ave of an operand."""
    RANK = {"grad": +1, "div": -1, "lap": 0, "skew": 0, "mulcos": 0, "ave": 0}

    def __init__(self, kind, arg):
        if not isinstance(arg, SOperand):
            raise ValueError("%s needs a field operand" % kind)
        if arg.basis is None:
            raise ValueError("%s of a constant" % kind)
        self.kind, self.arg, self.args = kind, arg, (arg,)
        self.dist, self.basis = arg.dist, arg.basis
        self.rank = arg.rank + self.RANK[kind]
        if self.rank < 0:
            raise ValueError("div needs a tensor of rank >= 1")
        self._dev = None

    def termlist(self, basis=None):
        return op_termlist(self.kind, basis or self.basis, self.arg.rank)

    def eval_c(self):
        ex = _ex(self.dist)
        if self._dev is None or self._dev[0] is not ex:
            tl = self.termlist()
            self._dev = (ex, ex.make_sphere_terms(self.basis.nm, self.basis.nl, tl.ncomp_out, tl.terms))
        x = self.arg.eval_c()
        y = ex.empty((self.ncomp, 2 * self.basis.nm, self.basis.nl))
        self._dev[1].apply(x, y)
        return y

    def lin(self, variables, basis):
        d, dt = self.arg.lin(variables, basis)
        tl = self.termlist(basis)
        return {i: tl.compose(t) for i, t in d.items()}, dt


class SAverage(SLinear):
    """ave(f): a constant (the reference's output basis is the 1 x 1 sphere, core/basis.py:5296-5320); kept here in the
    full coefficient layout with only (m, ell) = (0, 0) set."""

    def __init__(self, arg):
        SLinear.__init__(self, "ave", arg)
        self.is_constant = True


class SDt(SOperand):
    def __init__(self, arg):
        self.arg, self.args = arg, (arg,)
        self.dist, self.basis, self.rank = arg.dist, arg.basis, arg.rank

    def has_dt(self):
        return True

    def lin(self, variables, basis):
        d, dt = self.arg.lin(variables, basis)
        if dt:
            raise NonlinearError("nested time derivatives")
        return d, True

    def eval_c(self):
        raise ValueError("dt() cannot be evaluated")


def _bilinear_terms(rank_a, rank_b, contract):
    """(ic, ia, ib, coef) of the grid-space product of coordinate components: outer product (contract False) or
    contraction of the last index of a with the first of b (DotProduct, core/arithmetic.py:246-251, 855-866)."""
    ia_list = list(np.ndindex(*((2,) * rank_a)))
    ib_list = list(np.ndindex(*((2,) * rank_b)))
    if contract:
        out_list = list(np.ndindex(*((2,) * (rank_a + rank_b - 2))))
    else:
        out_list = list(np.ndindex(*((2,) * (rank_a + rank_b))))
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


class SProduct(SOperand):
    """Pointwise product on the dealiased grid (MultiplyFields, core/arithmetic.py:666-674)."""
    contract = False

    def __init__(self, a, b):
        self.args = (a, b)
        self.dist = a.dist
        self.basis = _basis_of(a, b)
        if a.basis is None or b.basis is None:
            raise NotImplementedError("product with a constant field")
        if self.contract:
            if a.rank < 1 or b.rank < 1:
                raise ValueError("dot product needs tensors of rank >= 1")
            self.rank = a.rank + b.rank - 2
        else:
            self.rank = a.rank + b.rank

    def eval_g(self):
        ex = _ex(self.dist)
        a, b = self.args
        ga, gb = a.eval_g(), b.eval_g()
        terms, nout = _bilinear_terms(a.rank, b.rank, self.contract)
        Np, Nt = self.basis.grid_shape(self.basis.dealias)
        out = ex.empty((nout, Np, Nt))
        ex.bilinear(out, nout, ga, gb, Np * Nt, terms)
        return out

    def lin(self, variables, basis):
        raise NonlinearError("products of fields are nonlinear")


class SDot(SProduct):
    contract = True


# ---- user-facing operator functions ------------------------------------------------------------------------------

def grad(a):
    return SLinear("grad", a)


def div(a):
    return SLinear("div", a)


def lap(a):
    return SLinear("lap", a)


def skew(a):
    return SLinear("skew", a)


def MulCosine(a):
    if isinstance(a, numbers.Number) and a == 0:
        return 0
    return SLinear("mulcos", a)


def ave(a, *coords):
    return SAverage(a)


def dt(a):
    return SDt(a)


# ==================================================================================================
# problems and solvers
# ==================================================================================================

def _split_equation(eq):
    from .problems import _split_equation as f
    return f(eq)


class SphereProblem:
    def __init__(self, variables, namespace=None, time="t"):
        self.variables = list(variables)
        self.dist = self.variables[0].dist
        self.basis = _basis_of(*self.variables)
        self.equations = []
        self.namespace = dict(grad=grad, div=div, lap=lap, skew=skew, MulCosine=MulCosine, ave=ave, dt=dt,
                              Gradient=grad, Divergence=div, Laplacian=lap, Skew=skew, Average=ave,
                              TimeDerivative=dt, np=np, numpy=np)
        if namespace:
            self.namespace.update({k: v for k, v in namespace.items()})
        for v in self.variables:
            if v.name:
                self.namespace[v.name] = v

    def _parse(self, side):
        if isinstance(side, (SOperand, numbers.Number)):
            return side
        return eval(side, dict(self.namespace))

    def add_equation(self, equation, condition=None):
        if isinstance(equation, str):
            lhs_s, rhs_s = _split_equation(equation)
            lhs, rhs = self._parse(lhs_s), self._parse(rhs_s)
        else:
            lhs, rhs = [self._parse(s) for s in equation]
        if not isinstance(lhs, SOperand):
            raise ValueError("LHS must involve the problem variables")
        if isinstance(rhs, SOperand) and rhs.has_dt():
            raise ValueError("time derivatives must be on the LHS")
        M, L = self._linearize(lhs)
        if isinstance(rhs, numbers.Number):
            if rhs != 0:
                raise NotImplementedError("non-zero constant right-hand sides on the sphere")
            F = None
        else:
            if rhs.rank != lhs.rank:
                raise ValueError("LHS and RHS tensor signatures differ")
            F = rhs
        constant = bool(getattr(lhs, "is_constant", False))
        eq = dict(lhs=lhs, rank=lhs.rank, ncomp=lhs.ncomp, M=M, L=L, F=F, constant=constant,
                  string=equation if isinstance(equation, str) else None)
        self.equations.append(eq)
        return eq

    def _linearize(self, lhs):
        """Split the LHS sum into dt-terms (M) and the rest (L): {variable index: TermList} each."""
        terms = []

        def flatten(node, scale):
            if isinstance(node, SAdd):
                for a in node.args:
                    flatten(a, scale)
            elif isinstance(node, SScale):
                flatten(node.arg, scale * node.a)
            else:
                terms.append((scale, node))
        flatten(lhs, 1.0)
        M, L = {}, {}
        for scale, node in terms:
            try:
                d, isdt = node.lin(self.variables, self.basis)
            except NonlinearError as e:
                raise ValueError("LHS must be linear in the problem variables: %s" % e)
            tgt = M if isdt else L
            for i, tl in d.items():
                tl = tl.scaled(scale)
                tgt[i] = tgt[i] + tl if i in tgt else tl
        return M, L

    def build_solver(self, *args, **kw):
        return self.solver_class(self, *args, **kw)


class SphereIVP(SphereProblem):
    @property
    def solver_class(self):
        return SphereInitialValueSolver


class SphereLBVP(SphereProblem):
    @property
    def solver_class(self):
        return SphereBoundaryValueSolver


class _SpherePack:
    """matvec interface the shared timesteppers use (pack.matvec(id, x, y))."""

    def __init__(self, solver):
        self.solver = solver
        self.mats = []

    def add(self, dev_terms):
        self.mats.append(dev_terms)
        return len(self.mats) - 1

    def matvec(self, mid, x, y):
        self.mats[mid].apply(x, y)


class SphereSolverBase:
    def __init__(self, problem):
        self.problem = problem
        self.dist = problem.dist
        self.ex = self.dist.executor
        self.basis = basis = problem.basis
        self.variables = problem.variables
        nm, nl = basis.nm, basis.nl
        # column layout: variables concatenated by component
        self.col0, c = [], 0
        for v in self.variables:
            self.col0.append(c)
            c += v.ncomp
        self.R = c
        self.row0, r = [], 0
        for eq in problem.equations:
            self.row0.append(r)
            r += eq["ncomp"]
        if r != c:
            raise ValueError("the problem is not square: %d equation components for %d variable components" % (r, c))
        self.nx, self.ny = 2 * nm, nl
        # validity of columns (variables) and rows (equations)
        self.col_valid = np.zeros((self.R, nm, nl), dtype=bool)
        for v, c0 in zip(self.variables, self.col0):
            if v.basis is None:
                self.col_valid[c0, 0, 0] = True
            else:
                self.col_valid[c0:c0 + v.ncomp] = basis.valid(v.rank)
        self.row_valid = np.zeros((self.R, nm, nl), dtype=bool)
        for eq, r0 in zip(problem.equations, self.row0):
            if eq["constant"]:
                self.row_valid[r0, 0, 0] = True
            else:
                self.row_valid[r0:r0 + eq["ncomp"]] = basis.valid(eq["rank"])
        # system term lists
        self.M_tl = self._system_termlist("M")
        self.L_tl = self._system_termlist("L")
        self.pack = _SpherePack(self)
        self.M_id = self.pack.add(self.ex.make_sphere_terms(nm, nl, self.R, self.M_tl.terms))
        self.L_id = self.pack.add(self.ex.make_sphere_terms(nm, nl, self.R, self.L_tl.terms))
        # state vector; variable coefficient arrays are views into it
        self.X = self.ex.zeros((self.R, self.nx, self.ny))
        self._inverses = {}
        self._adopt_state()

    def _system_termlist(self, which):
        tl = TermList(self.R, self.R, [])
        for eq, r0 in zip(self.problem.equations, self.row0):
            for i, t in eq[which].items():
                tl.terms += t.embed(r0, self.col0[i], self.R, self.R).terms
        tl = tl.merged()
        # rows and columns of invalid modes carry nothing
        out = []
        for (co, ci, d, cf) in tl.terms:
            nl = cf.shape[1]
            colv = np.zeros_like(self.col_valid[ci])
            if d >= 0:
                colv[:, :nl - d] = self.col_valid[ci][:, d:]
            else:
                colv[:, -d:] = self.col_valid[ci][:, :nl + d]
            out.append((co, ci, d, cf * self.row_valid[co] * colv))
        return TermList(self.R, self.R, out).merged()

    def _adopt_state(self):
        self._views = []
        for v, c0 in zip(self.variables, self.col0):
            view = self.X[c0:c0 + v.ncomp]
            self._views.append(view)
            if v.basis is None:
                continue
            self.ex.copy(view, v.require_coeff_space())
            v._set_device_coeff(view)

    def sync_state_to_device(self):
        """Variables the user (or a transform) moved out of the state vector are copied back in."""
        for v, view in zip(self.variables, self._views):
            if v.basis is None:
                continue
            c = v.require_coeff_space()
            if c is not view:
                self.ex.copy(view, c)
                v._set_device_coeff(view)

    def mark_state_current(self):
        for v, view in zip(self.variables, self._views):
            if v.basis is None:
                continue
            v._set_device_coeff(view)
            v._g = None

    # ---- per-m dense matrices --------------------------------------------------------------------------------------
    def _dense(self, tl, m):
        """Dense complex matrix of a system term list at azimuthal wavenumber m, unknown j = comp * ne + (ell - m)."""
        nl = self.basis.nl
        ne = nl - m
        A = np.zeros((self.R * ne, self.R * ne), dtype=complex)
        if ne <= 0:
            return A
        ell = np.arange(m, nl)
        for (co, ci, d, cf) in tl.terms:
            src = ell + d
            ok = (src >= m) & (src < nl)
            A[co * ne + (ell[ok] - m), ci * ne + (src[ok] - m)] += cf[m, ell[ok]]
        return A

    _dinv = None

    # ---- inverses through the band LU -----------------------------------------------------------------------------------
    # A sphere system has ONE right-hand side per step, so its solve stays a dense-inverse GEMV (a banded sweep of 765
    # dependent rows cannot be hidden behind 256 systems: measured in round 2).  But the inverse itself need not cost
    # O(n^3): the per-m matrices are block-tridiagonal in ell, so  (a M + b L)_m^-1  is obtained row by row from unit
    # solves against the band LU of the transposed system -- the machinery of the shell (core/ellband.py,
    # csrc/ddh_ellband.hip), fed with the REAL form of the complex matrices (components 2c / 2c + 1 = re / im).
    _sband = None

    def _real_form_transposed(self, tl, m):
        """scipy.sparse real form, index (2c + part) nl + ell, of the TRANSPOSE of the complex term-list matrix at m"""
        from scipy import sparse
        nl = self.basis.nl
        ell = np.arange(m, nl)
        rows, cols, vals = [], [], []
        for (co, ci, d, cf) in tl.terms:
            src = ell + d
            ok = (src >= m) & (src < nl)
            v = cf[m, ell[ok]]
            # A[(co, l), (ci, l + d)] = v  ->  A^T[(ci, l + d), (co, l)] = v;  z -> v z in real form: [[vr, -vi], [vi, vr]]
            r0, r1 = (2 * ci) * nl + src[ok], (2 * ci + 1) * nl + src[ok]
            c0, c1 = (2 * co) * nl + ell[ok], (2 * co + 1) * nl + ell[ok]
            rows += [r0, r0, r1, r1]
            cols += [c0, c1, c0, c1]
            vals += [v.real, -v.imag, v.imag, v.real]
        n = 2 * self.R * nl
        if not rows:
            return sparse.csr_matrix((n, n))
        return sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))

    def _band_inverse_setup(self):
        """-> dict of the band path, or False (executor without it, DDH_SPHERE_DENSE=1, matrices that are no bands)"""
        if self._sband is None:
            self._sband = False
            import os
            nm, nl, R = self.basis.nm, self.basis.nl, self.R
            if hasattr(self.ex, "make_ell_band") and os.environ.get("DDH_SPHERE_DENSE", "0") != "1" and \
                    getattr(self.dist, "size", 1) == 1:
                from .ellband import EllBandPlan
                above = lambda m: (np.arange(nl) >= m)[None, :]
                rv = [np.repeat(self.row_valid[:, m, :] * above(m), 2, axis=0) for m in range(nm)]
                cv = [np.repeat(self.col_valid[:, m, :] * above(m), 2, axis=0) for m in range(nm)]
                # (transposed systems: equation modes are the columns)
                plan = EllBandPlan(lambda m: self._real_form_transposed(self.M_tl, m),
                                   lambda m: self._real_form_transposed(self.L_tl, m), cv, rv, [], [], nl, range(nm))
                if plan.per and not plan.dense_groups and plan.nbc == 0 and plan.mp == 0:
                    nslots = R * nl
                    limit = [R * max(nl - m, 0) for m in range(nm)]          # slots (nl - 1 - ell) R + c with ell >= m
                    dev = self.ex.make_ell_band(plan, 2 * R, nslots, nm, nl, limit)
                    rhs = np.zeros((2 * R, nslots, nm, nl))
                    for m in range(nm):
                        for c in range(R):
                            ell = np.flatnonzero(self.col_valid[c, m, :] * (np.arange(nl) >= m))     # (unknown (c, ell): a row of A^T)
                            rhs[2 * c, (nl - 1 - ell) * R + c, m, ell] = 1.0
                    off = np.concatenate([[0], np.cumsum([(R * max(nl - m, 0)) ** 2 for m in range(nm)])]).astype(np.int64)
                    self._sband = dict(plan=plan, dev=dev, rhs=self.ex.from_host(rhs), x=self.ex.zeros((2 * R, nslots, nm, nl)),
                                       off=self.ex.from_host_int64(off), count=int(off[-1]), nslots=nslots)
                    logger.info("sphere LHS: inverses from the band LU of the real-form systems (kl %d, ku %d, %d unknowns at m = 0)"
                                % (plan.kl, plan.ku, plan.nmax))
        return self._sband

    def _inverse_batch(self, a, b, old=None):
        """Per-m inverse of (a M + b L) restricted to the valid modes, embedded with zeros elsewhere, applied as one
        batched complex GEMV.  Formed on the device: from unit solves against the band LU of the transposed real-form
        systems (see above) where the matrices allow it, else by the dense Gauss-Jordan kernel
        (executor.make_dense_inverse; M_m, L_m are uploaded once)."""
        nm, nl = self.basis.nm, self.basis.nl
        band = self._band_inverse_setup()
        if band:
            band["dev"].factor(a, b, index=0)
            band["dev"].solve(0, band["rhs"], band["x"])
            flat = self.ex.gather_complex_inverse(band["x"], band["off"], band["count"], self.R, nl, nm, band["nslots"])
            return self.ex.make_cgemv_batch_flat(nm, nl, self.R, flat, old=old)
        if self._dinv is None:
            Ms, Ls, rvs, cvs = [], [], [], []
            for m in range(nm):
                ne = max(nl - m, 0)
                n = self.R * ne
                if ne > 0:
                    Ms.append(self._dense(self.M_tl, m))
                    Ls.append(self._dense(self.L_tl, m))
                    rv = self.row_valid[:, m, m:].reshape(-1)
                    cv = self.col_valid[:, m, m:].reshape(-1)
                    if rv.sum() != cv.sum():
                        raise ValueError("m = %d: %d valid equation modes for %d valid variable modes" % (m, rv.sum(), cv.sum()))
                else:
                    Ms.append(np.zeros((0, 0), dtype=complex))
                    Ls.append(np.zeros((0, 0), dtype=complex))
                    rv = cv = np.zeros(0, dtype=bool)
                rvs.append(rv)
                cvs.append(cv)
            self._dinv = self.ex.make_dense_inverse(Ms, Ls, rvs, cvs, complex_=True)
        return self.ex.make_cgemv_batch_flat(nm, nl, self.R, self._dinv.compute(a, b), old=old)

    def evaluate_F(self, out, persistent=False):
        ex = self.ex
        ex.fill_zero(out)
        self.dist._grid_cache = {}
        try:
            for eq, r0 in zip(self.problem.equations, self.row0):
                if eq["F"] is None:
                    continue
                c = eq["F"].eval_c()
                ex.copy(out[r0:r0 + eq["ncomp"]], c)
        finally:
            self.dist._grid_cache = None


class SphereBoundaryValueSolver(SphereSolverBase):
    """L.X = F  (core/solvers.py:LinearBoundaryValueSolver): one batched per-m solve."""

    def __init__(self, problem, **kw):
        super().__init__(problem)
        if self.M_tl.terms:
            raise ValueError("LBVP equations cannot contain time derivatives")
        self._inv = None

    def solve(self):
        if self._inv is None:
            self._inv = self._inverse_batch(0.0, 1.0)
        F = self.ex.zeros((self.R, self.nx, self.ny))
        self.evaluate_F(F)
        self._inv.apply(F, self.X)
        self.mark_state_current()
        # constants: read their value back
        for v, c0 in zip(self.variables, self.col0):
            if v.basis is None:
                val = np.asarray(self.ex.download(self.X[c0:c0 + 1]))[0, 0, 0]
                v._const = np.array(val * SphereBasis.constant_mode_value)
                v._authority = "device"


class SphereInitialValueSolver(IVPLifecycle, SphereSolverBase):
    """IMEX timestepping of M.dt(X) + L.X = F on the sphere (core/solvers.py:InitialValueSolver); the schemes are
    the shared ones of core/timesteppers.py, the life cycle is the shared core/ivp_common.py::IVPLifecycle."""

    def __init__(self, problem, timestepper, enforce_real_cadence=100, warmup_iterations=10, **kw):
        t0 = _time.time()
        SphereSolverBase.__init__(self, problem)
        from . import timesteppers as ts
        if isinstance(timestepper, str):
            timestepper = ts.schemes[timestepper]
        self.sim_time = self.initial_sim_time = 0.0
        self._init_lifecycle(enforce_real_cadence, warmup_iterations)
        self._lus = []
        self.timestepper = timestepper(self)
        self.setup_time = _time.time() - t0
        self.total_modes = int(self.col_valid.sum()) * 2
        from .output import OutputEvaluator
        self.evaluator = OutputEvaluator(self)       # analysis handlers: evaluated at the start of a step
        self._step_hooks = [self.evaluator.step_hook]

    @property
    def state(self):
        return self.variables

    def _hermitian_round_trip(self, f):
        if isinstance(f, SField) and f.basis is not None:
            f.require_grid_space(f.basis.dealias)
            f.require_coeff_space()

    # interface used by the shared timesteppers ------------------------------------------------------------------
    def factor(self, a, b, reuse=-1):
        prev = self._lus[reuse] if (reuse is not None and 0 <= reuse < len(self._lus)) else None
        inv = self._inverse_batch(a, b, old=prev)
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
        self._lus[lu].apply(rhs, x)
        probe = getattr(self, "solve_probe", None)
        if probe is not None:                        # parity checks: keep (a, b, rhs, x) of every solve
            a, b = self._lu_params[lu]
            probe.append(dict(a=a, b=b, rhs=self.ex.download(rhs).copy(), x=self.ex.download(x).copy()))

