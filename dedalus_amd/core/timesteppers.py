"""
IMEX timesteppers on device-resident system vectors.

Same schemes, names and stage algebra as dedalus/core/timesteppers.py:
  multistep (:34-187)   (a0 M + b0 L) X_n = sum_j c_j F_{n-j} - sum_j a_j M X_{n-j} - sum_j b_j L X_{n-j}
  Runge-Kutta (:498-644) (M + k H_ii L) X_i = M X_0 + k sum_j A_ij F_j - k sum_j H_ij L X_j
but every per-subproblem Python loop is one batched kernel over all pencils:
  M.X / L.X  -> ddh_pencil_matvec,   RHS assembly -> ddh_lincomb,   LHS solve -> ddh_pencil_solve,
  LHS refactorisation (when a0,b0 or k*H_ii change) -> ddh_pencil_factor.

The variable-step multistep coefficients are not tabulated: they are computed from their defining
conditions (finite-difference / extrapolation / Adams weights on the actual time nodes), which
reproduces the closed forms of Wang & Ruuth (2008) used by the reference to round-off
(pinned by tests/test_ivp_oracle.py against reference-generated fixtures, tests/golden/timesteppers.npz).
"""

import math
from collections import deque

import numpy as np

schemes = {}


def add_scheme(cls):
    schemes[cls.__name__] = cls
    return cls


def _fd_weights(nodes, x0, deriv):
    """Weights w_j with sum_j w_j f(nodes_j) = f^(deriv)(x0), exact for polynomials of degree len-1."""
    nodes = np.asarray(nodes, dtype=np.longdouble) - np.longdouble(x0)
    n = len(nodes)
    V = np.vander(nodes, n, increasing=True).T          # V[p, j] = nodes_j^p
    rhs = np.zeros(n, dtype=np.longdouble)
    rhs[deriv] = math.factorial(deriv)
    return _solve_ld(V, rhs)


def _solve_ld(A, b):
    """Gaussian elimination with partial pivoting in long double (tiny systems)."""
    A = np.array(A, dtype=np.longdouble)
    b = np.array(b, dtype=np.longdouble)
    n = len(b)
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
            b[[k, p]] = b[[p, k]]
        for i in range(k + 1, n):
            m = A[i, k] / A[k, k]
            A[i, k:] -= m * A[k, k:]
            b[i] -= m * b[k]
    x = np.zeros(n, dtype=np.longdouble)
    for k in range(n - 1, -1, -1):
        x[k] = (b[k] - A[k, k + 1:] @ x[k + 1:]) / A[k, k]
    return x.astype(np.float64)


def _times(timesteps, n):
    """Time nodes t_n = 0, t_{n-1} = -k_n, t_{n-2} = -k_n - k_{n-1}, ... (n+1 nodes)."""
    t = [0.0]
    for j in range(n):
        t.append(t[-1] - timesteps[j])
    return t


# ==================================================================================================
# multistep
# ==================================================================================================

class _SolveMixin:
    """RHS = sum_t alpha_t x_t followed by the LHS solve: solvers with a fused path (SolverBase.solve_lincomb, the band
    engine's forward sweep forms the combination on the fly) never materialise the RHS; the others get the axpy chain
    into a RHS buffer (timesteppers.py:156-166, 617-623) and a plain solve."""
    _RHS = None

    @property
    def RHS(self):
        if self._RHS is None:
            s = self.solver
            self._RHS = s.ex.zeros((s.R, s.nx, s.ny))
        return self._RHS

    def _solve_combination(self, xs, al, lu, zero_rows=None, skip_rows=None, tiled=False):
        s = self.solver
        if hasattr(s, "solve_lincomb"):
            if tiled:
                s.solve_lincomb(lu, xs, al, s.X, zero_rows=zero_rows, skip_rows=skip_rows, tiled=True)
            elif zero_rows is not None or skip_rows is not None:
                s.solve_lincomb(lu, xs, al, s.X, zero_rows=zero_rows, skip_rows=skip_rows)
            else:
                s.solve_lincomb(lu, xs, al, s.X)
        else:
            s.ex.lincomb(self.RHS, xs, al)
            s.solve(lu, self.RHS, s.X)


class MultistepIMEX(_SolveMixin):
    stages = 1

    def __init__(self, solver):
        self.solver = solver
        ex, shape = solver.ex, (solver.R, solver.nx, solver.ny)
        self.dt = deque([0.0] * self.steps)
        self.MX = deque(ex.zeros(shape) for _ in range(self.amax))
        self.F = deque(ex.zeros(shape) for _ in range(self.cmax))
        if hasattr(solver, "_F_zeroed"):
            solver._F_zeroed = set()            # (addresses of an earlier timestepper's buffers may be handed out again)
        self._iteration = 0
        self._LHS_params = None
        self._lu = -1
        # L.X history only where the scheme's explicit-in-L weights b_1.. are not identically zero (the SBDF family
        # never reads it: skip the product)
        self._need_lx = False
        for it in range(self.steps + 1):
            _, b, _ = self.compute_coefficients([1.0] * self.steps, it)
            self._need_lx = self._need_lx or bool(np.any(np.asarray(b)[1:] != 0.0))
        self.LX = deque((ex.zeros(shape) if self._need_lx else None) for _ in range(self.bmax))

    # ---- fixed-timestep steps as HIP graphs (core/ivp_common.py::_graph_replay) -------------------------------------------
    # A multistep step is the same launch sequence every time once the start-up orders are over and the timestep history
    # is constant -- except that the history buffers rotate: the pattern of buffer addresses repeats with the period
    # below, so one graph per phase is captured and the phases are replayed in turn.
    def graph_phase(self, dt):
        """-> (period, phase) of the step about to be taken, or None while the step still differs from its successors"""
        if self._iteration < self.steps + 1 or any(h != dt for h in self.dt):
            return None
        period = int(np.lcm.reduce([len(self.MX), len(self.LX), len(self.F)]))
        return period, self._iteration % period

    def graph_snapshot(self):
        """host state a step mutates (taken before a capture: the capture runs the host side of step() without
        executing a kernel, so a capture that fails must be undone before the step is taken by ordinary launches)"""
        return (list(self.dt), self._iteration, list(self.MX), list(self.LX), list(self.F), self._LHS_params, self._lu)

    def graph_rollback(self, snap):
        dts, self._iteration, mx, lx, f, self._LHS_params, self._lu = snap
        self.dt, self.MX, self.LX, self.F = deque(dts), deque(mx), deque(lx), deque(f)

    def graph_advance(self, dt):
        """the host-side part of a step whose launches were replayed from a graph"""
        self.dt.rotate()
        self.dt[0] = dt
        self._iteration += 1
        self.MX.rotate()
        self.LX.rotate()
        self.F.rotate()

    def step(self, dt, wall_time=None):
        s = self.solver
        ex, pack = s.ex, s.pack
        self.dt.rotate()
        self.dt[0] = dt
        a, b, c = self.compute_coefficients(list(self.dt), self._iteration)
        self._iteration += 1
        self.MX.rotate()
        self.LX.rotate()
        self.F.rotate()
        s.sync_state_to_device()
        pack.matvec(s.M_id, s.X, self.MX[0])
        if self._need_lx:
            pack.matvec(s.L_id, s.X, self.LX[0])
        s.evaluate_F(self.F[0], persistent=True)
        xs, al = [], []
        for j in range(1, len(c)):
            if c[j] != 0.0:
                xs.append(self.F[j - 1]); al.append(c[j])
        for j in range(1, len(a)):
            if a[j] != 0.0:
                xs.append(self.MX[j - 1]); al.append(-a[j])
        for j in range(1, len(b)):
            if b[j] != 0.0:
                xs.append(self.LX[j - 1]); al.append(-b[j])
        if (a[0], b[0]) != self._LHS_params:
            self._lu = s.factor(a[0], b[0], reuse=self._lu)
            self._LHS_params = (a[0], b[0])
        self._solve_combination(xs, al, self._lu)
        s.mark_state_current()
        s.sim_time = s.sim_time + dt


def _pad(v, n):
    out = np.zeros(n + 1)
    out[:len(v)] = v
    return out


@add_scheme
class CNAB1(MultistepIMEX):
    """Crank-Nicolson / forward Euler [Wang & Ruuth 2008 eq 2.5.3]"""
    amax = bmax = cmax = steps = 1

    @classmethod
    def compute_coefficients(cls, timesteps, iteration):
        k0 = timesteps[0]
        a = _fd_weights(_times(timesteps, 1), 0.0, 1)           # two-point derivative
        return _pad(a, cls.amax), _pad([0.5, 0.5], cls.bmax), _pad([0.0, 1.0], cls.cmax)


@add_scheme
class SBDF1(MultistepIMEX):
    """Backward Euler / forward Euler [Wang & Ruuth 2008 eq 2.6]"""
    amax = bmax = cmax = steps = 1

    @classmethod
    def compute_coefficients(cls, timesteps, iteration):
        a = _fd_weights(_times(timesteps, 1), 0.0, 1)
        return _pad(a, cls.amax), _pad([1.0], cls.bmax), _pad([0.0, 1.0], cls.cmax)


def _adams_bashforth(timesteps, order):
    """c_j (j=1..order): F integrated over the last step from the interpolant through t_{n-1..n-order},
    normalised by the step (variable-step Adams-Bashforth)."""
    t = _times(timesteps, order)           # t[0]=t_n, t[1]=t_{n-1}, ...
    nodes = np.array(t[1:order + 1], dtype=np.longdouble)
    k = np.longdouble(timesteps[0])
    # weights w_j with sum_j w_j p(nodes_j) = (1/k) int_{t_{n-1}}^{t_n} p, exact for degree < order
    V = np.vander(nodes, order, increasing=True).T
    p = np.arange(order, dtype=np.longdouble)
    rhs = (np.longdouble(t[0]) ** (p + 1) - np.longdouble(t[1]) ** (p + 1)) / (p + 1) / k
    return np.concatenate([[0.0], _solve_ld(V, rhs)])


def _extrapolation(timesteps, order):
    """c_j (j=1..order): polynomial extrapolation of F from t_{n-1..n-order} to t_n."""
    t = _times(timesteps, order)
    w = _fd_weights(t[1:order + 1], t[0], 0)
    return np.concatenate([[0.0], w])


@add_scheme
class CNAB2(MultistepIMEX):
    """Crank-Nicolson / 2nd-order Adams-Bashforth [Wang & Ruuth 2008 eq 2.9]"""
    amax = bmax = cmax = steps = 2

    @classmethod
    def compute_coefficients(cls, timesteps, iteration):
        if iteration < 1:
            a, b, c = CNAB1.compute_coefficients(timesteps, iteration)
            return _pad(a, 2), _pad(b, 2), _pad(c, 2)
        a = _fd_weights(_times(timesteps, 1), 0.0, 1)
        return _pad(a, 2), _pad([0.5, 0.5], 2), _pad(_adams_bashforth(timesteps, 2), 2)


@add_scheme
class MCNAB2(MultistepIMEX):
    """Modified Crank-Nicolson / 2nd-order Adams-Bashforth [Wang & Ruuth 2008 eq 2.10]"""
    amax = bmax = cmax = steps = 2

    @classmethod
    def compute_coefficients(cls, timesteps, iteration):
        if iteration < 1:
            a, b, c = CNAB1.compute_coefficients(timesteps, iteration)
            return _pad(a, 2), _pad(b, 2), _pad(c, 2)
        w = timesteps[0] / timesteps[1]
        a = _fd_weights(_times(timesteps, 1), 0.0, 1)
        # the modified implicit weights are the definition of the scheme (Wang & Ruuth eq 2.10)
        b = [(8 + 1 / w) / 16, (7 - 1 / w) / 16, 1 / 16]
        return _pad(a, 2), _pad(b, 2), _pad(_adams_bashforth(timesteps, 2), 2)


class _SBDF(MultistepIMEX):
    @classmethod
    def _coeffs(cls, timesteps, order, size):
        a = _fd_weights(_times(timesteps, order), 0.0, 1)       # BDF derivative at t_n
        return _pad(a, size), _pad([1.0], size), _pad(_extrapolation(timesteps, order), size)


@add_scheme
class SBDF2(_SBDF):
    """2nd-order semi-implicit BDF [Wang & Ruuth 2008 eq 2.8]"""
    amax = bmax = cmax = steps = 2

    @classmethod
    def compute_coefficients(cls, timesteps, iteration):
        order = min(2, iteration + 1)
        return cls._coeffs(timesteps, order, 2)


@add_scheme
class CNLF2(MultistepIMEX):
    """Crank-Nicolson leap-frog [Wang & Ruuth 2008 eq 2.11]"""
    amax = bmax = cmax = steps = 2

    @classmethod
    def compute_coefficients(cls, timesteps, iteration):
        if iteration < 1:
            a, b, c = CNAB1.compute_coefficients(timesteps, iteration)
            return _pad(a, 2), _pad(b, 2), _pad(c, 2)
        t = _times(timesteps, 2)
        w = timesteps[0] / timesteps[1]
        a = _fd_weights(t, t[1], 1)                              # centred derivative at t_{n-1}
        b = [1 / w / 2, (1 - 1 / w) / 2, 1 / 2]                  # scheme definition (eq 2.11)
        return _pad(a, 2), _pad(b, 2), _pad([0.0, 1.0], 2)


@add_scheme
class SBDF3(_SBDF):
    """3rd-order semi-implicit BDF [Wang & Ruuth 2008 eq 2.14]"""
    amax = bmax = cmax = steps = 3

    @classmethod
    def compute_coefficients(cls, timesteps, iteration):
        order = min(3, iteration + 1)
        return cls._coeffs(timesteps, order, 3)


@add_scheme
class SBDF4(_SBDF):
    """4th-order semi-implicit BDF [Wang & Ruuth 2008 eq 2.15]"""
    amax = bmax = cmax = steps = 4

    @classmethod
    def compute_coefficients(cls, timesteps, iteration):
        order = min(4, iteration + 1)
        return cls._coeffs(timesteps, order, 4)


# ==================================================================================================
# Runge-Kutta
# ==================================================================================================

class RungeKuttaIMEX(_SolveMixin):
    steps = 1

    def __init__(self, solver):
        import os
        self.solver = solver
        ex, shape = solver.ex, (solver.R, solver.nx, solver.ny)
        H = self.H
        self.MX0 = ex.zeros(shape)
        self.F = [ex.zeros(shape) for _ in range(self.stages)]
        if hasattr(solver, "_F_zeroed"):
            solver._F_zeroed = set()            # (addresses of an earlier timestepper's buffers may be handed out again)
        # Which L.X_j enter a later stage?  (the first column of H is zero in the registered schemes)
        self._need_lx = [any(H[m, j] != 0.0 for m in range(j + 1, self.stages + 1)) for j in range(self.stages + 1)]
        # L.X_j of a solved stage j >= 1 is not formed by a mat-vec: the stage equation (M + k H_jj L) X_j = RHS_j
        # gives L.X_j = (RHS_j - M.X_j) / (k H_jj) exactly, and M has ~1/3 of L's entries (and none of its
        # wavenumber-dependent terms).  The term enters later right-hand sides multiplied by k H_ij, so the solve
        # residual hidden in the identity is scaled by H_ij / H_jj = O(1): round-off level, like any reordering of the
        # sums.  DDH_RK_DIRECT_LX=1 restores the explicit products (reference order of operations,
        # core/timesteppers.py:588-604).
        self._direct = os.environ.get("DDH_RK_DIRECT_LX", "0") == "1"
        self.LX = [ex.zeros(shape) if (self._need_lx[j] and (self._direct or j == 0)) else None
                   for j in range(self.stages + 1)]
        self.MX = [None] + [ex.zeros(shape) if (self._need_lx[j] and not self._direct) else None
                            for j in range(1, self.stages + 1)]
        self._k = None
        self._lus = {}          # H_ii -> lu id

    def step(self, dt, wall_time=None):
        s = self.solver
        ex, pack = s.ex, s.pack
        k = dt
        A, H, c = self.A, self.H, self.c
        if k != self._k:
            # refactor once per distinct diagonal coefficient (the reference refactors per stage)
            old = self._lus
            self._lus = {}
            reuse = list(old.values())
            for i in range(1, self.stages + 1):
                h = float(H[i, i])
                if h not in self._lus:
                    self._lus[h] = s.factor(1.0, k * h, reuse=(reuse.pop() if reuse else -1))
            self._k = k
        t0 = s.sim_time
        s.sync_state_to_device()
        own = dict(owned=True) if getattr(pack, "supports_zero_rows", False) else {}
        # the M.X and F buffers of this timestepper in the tile-major layout the sweeps read contiguously
        # (SolverBase.rhs_tiling).  The answer belongs to the current factorizations and sweep variant: asked again after
        # every refactorization and every PencilPack.set_solve_variant.  Every buffer is rewritten in every step, so a
        # change of layout only has to restore the zeros of the rows nothing writes.
        stamp = (tuple(self._lus.values()), getattr(pack, "variant_epoch", 0))
        if getattr(self, "_tiled_stamp", None) != stamp:
            tiled_now = 0
            if own and not self._direct and not any(v is not None for v in self.LX) and hasattr(s, "rhs_tiling"):
                tiled_now = s.rhs_tiling(self._lus.values())
            if getattr(self, "_tiled", None) is not None and bool(tiled_now) != bool(self._tiled):
                for buf in [self.MX0] + self.F + [v for v in self.MX if v is not None]:
                    ex.fill_zero(buf)
                s._F_zeroed = set(b.data_ptr() if hasattr(b, "data_ptr") else id(b) for b in self.F)
            self._tiled, self._tiled_stamp = tiled_now, stamp
        tiled = self._tiled
        if tiled:
            own = dict(owned=True, tiled=True)
        # The mat-vecs of a stage read the state only: a solver whose right-hand-side evaluation has a wait in it (several
        # ranks: the last window's forward exchange, SolverBase.evaluate_F) takes them as work to issue at that point.
        later = []
        defer_mv = bool(getattr(s, "overlaps_rhs_work", lambda: False)())
        (later.append if defer_mv else (lambda f: f()))(lambda: pack.matvec(s.M_id, s.X, self.MX0, **own))
        combs = {}
        for i in range(1, self.stages + 1):
            j = i - 1                                   # s.X holds X_j
            if self._need_lx[j]:
                if self._direct or j == 0:
                    mv = (lambda j=j: pack.matvec(s.L_id, s.X, self.LX[j]))
                else:
                    mv = (lambda j=j: pack.matvec(s.M_id, s.X, self.MX[j], **own))
                (later.append if defer_mv else (lambda f: f()))(mv)
            if later:
                s._overlap_work = list(later)
                del later[:]
            if tiled:
                s.evaluate_F(self.F[i - 1], persistent=True, tiled_row=tiled)
            else:
                s.evaluate_F(self.F[i - 1], persistent=True)
            # RHS_i as a combination of the stored vectors.  RHS_j of an earlier stage is itself such a combination
            # (it is not kept): -k H_ij L.X_j = -(H_ij / H_jj) (RHS_j - M.X_j) expands into MX0, F_*, MX_* terms.
            comb = {("MX0",): 1.0}
            for j in range(i):
                if A[i, j] != 0.0:
                    comb[("F", j)] = comb.get(("F", j), 0.0) + k * A[i, j]
                if H[i, j] != 0.0:
                    if self._direct or j == 0:
                        comb[("LX", j)] = comb.get(("LX", j), 0.0) - k * H[i, j]
                    else:
                        r = H[i, j] / H[j, j]
                        for key, v in combs[j].items():
                            comb[key] = comb.get(key, 0.0) - r * v
                        comb[("MX", j)] = comb.get(("MX", j), 0.0) + r
            combs[i] = comb
            vec = {"MX0": lambda q: self.MX0, "F": lambda q: self.F[q], "LX": lambda q: self.LX[q], "MX": lambda q: self.MX[q]}
            xs = [vec[key[0]](key[1] if len(key) > 1 else None) for key, v in comb.items() if v != 0.0]
            al = [v for v in comb.values() if v != 0.0]
            # a right-hand side made of M.X and F vectors only: rows that are structurally zero in both are not read
            zrows = None
            if not any(key[0] == "LX" and v != 0.0 for key, v in comb.items()) and hasattr(s, "mx_f_zero_rows"):
                zrows = s.mx_f_zero_rows()
            # an intermediate stage: unknowns that neither M.X nor F read (pressure, taus) are not stored -- they are final
            # only after the last stage (not with DDH_RK_DIRECT_LX: L.X_j reads every unknown)
            skip = None
            if i < self.stages and not self._direct and hasattr(s, "intermediate_skip_rows"):
                skip = s.intermediate_skip_rows()
            self._solve_combination(xs, al, self._lus[float(H[i, i])], zero_rows=zrows, skip_rows=skip, tiled=bool(tiled))
            s.mark_state_current()
            s.sim_time = t0 + k * c[i]


@add_scheme
class RK111(RungeKuttaIMEX):
    """1-stage 1st-order DIRK+ERK [Ascher, Ruuth & Spiteri 1997 sec 2.1]"""
    stages = 1
    c = np.array([0.0, 1.0])
    A = np.array([[0.0, 0.0], [1.0, 0.0]])
    H = np.array([[0.0, 0.0], [0.0, 1.0]])


@add_scheme
class RK222(RungeKuttaIMEX):
    """2-stage 2nd-order DIRK+ERK [Ascher, Ruuth & Spiteri 1997 sec 2.6]"""
    stages = 2
    _g = 1.0 - np.sqrt(0.5)
    _d = 1.0 - 1.0 / (2.0 * _g)
    c = np.array([0.0, _g, 1.0])
    A = np.array([[0.0, 0.0, 0.0], [_g, 0.0, 0.0], [_d, 1.0 - _d, 0.0]])
    H = np.array([[0.0, 0.0, 0.0], [0.0, _g, 0.0], [0.0, 1.0 - _g, _g]])


@add_scheme
class RK443(RungeKuttaIMEX):
    """4-stage 3rd-order DIRK+ERK [Ascher, Ruuth & Spiteri 1997 sec 2.8]"""
    stages = 4
    c = np.array([0.0, 1 / 2, 2 / 3, 1 / 2, 1.0])
    A = np.array([[0, 0, 0, 0, 0],
                  [1 / 2, 0, 0, 0, 0],
                  [11 / 18, 1 / 18, 0, 0, 0],
                  [5 / 6, -5 / 6, 1 / 2, 0, 0],
                  [1 / 4, 7 / 4, 3 / 4, -7 / 4, 0]], dtype=float)
    H = np.array([[0, 0, 0, 0, 0],
                  [0, 1 / 2, 0, 0, 0],
                  [0, 1 / 6, 1 / 2, 0, 0],
                  [0, -1 / 2, 1 / 2, 1 / 2, 0],
                  [0, 3 / 2, -3 / 2, 1 / 2, 1 / 2]], dtype=float)


@add_scheme
class RKSMR(RungeKuttaIMEX):
    """3-stage (3-eps)-order scheme of Spalart, Moser & Rogers (1991), Appendix"""
    stages = 3
    _al = (29 / 96, -3 / 40, 1 / 6)
    _be = (37 / 160, 5 / 24, 1 / 6)
    _ga = (8 / 15, 5 / 12, 3 / 4)
    _ze = (0.0, -17 / 60, -5 / 12)
    c = np.array([0.0, 8 / 15, 2 / 3, 1.0])
    A = np.zeros((4, 4))
    H = np.zeros((4, 4))
    for _i in range(1, 4):
        # explicit: stage i uses gamma_i on F_{i-1} and zeta_i on F_{i-2}, accumulated over stages
        for _j in range(1, _i + 1):
            A[_i, _j - 1] += _ga[_j - 1]
            if _j >= 2:
                A[_i, _j - 2] += _ze[_j - 1]
            H[_i, _j - 1] += _al[_j - 1]
            H[_i, _j] += _be[_j - 1]
    del _i, _j


class RKGFY(RungeKuttaIMEX):
    """2-stage 2nd-order scheme (defined but not registered in the reference either, timesteppers.py:727)"""
    stages = 2
    c = np.array([0.0, 1.0, 1.0])
    A = np.array([[0, 0, 0], [1, 0, 0], [0.5, 0.5, 0]], dtype=float)
    H = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5]], dtype=float)
