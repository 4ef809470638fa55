"""CFL and global flow properties (dedalus/extras/flow_tools.py:49-233).

The CFL frequency max_x sum_i |u_i| / dx_i is reduced on the device (ddh_grid_cfl) from the
dealias-scale grid velocity, with the reference's spacings (CartesianAdvectiveCFL,
core/basis.py:6078-6111) and the reference's scheduling: frequencies are sampled at the start of a step
whenever iteration % cadence == 0 (the dictionary handler of flow_tools.py:176) and consumed by
compute_timestep() one step later (flow_tools.py:183-207)."""

import numpy as np


class GlobalFlowProperty:
    """Reductions of grid-space expressions every `cadence` iterations (flow_tools.py:49-111)."""

    def __init__(self, solver, cadence=1):
        self.solver = solver
        self.cadence = cadence
        self.properties = {}
        self._cache = {}
        self._cache_iter = None
        self._sampled = {}
        # the reference evaluates the properties with a dictionary handler scheduled at iter % cadence == 0, i.e. at
        # the START of those steps (flow_tools.py:60-63, core/solvers.py evaluate_scheduled); max()/min()/... read the
        # last scheduled evaluation
        if hasattr(solver, "_step_hooks"):
            solver._step_hooks.append(self._sample)

    def add_property(self, property, name, precompute_integral=False):
        self.properties[name] = property

    def _evaluate(self, name):
        """(min, max, sum, count) of the property's grid data.  The reduction runs on the device (ddh_grid_reduce, the
        local part of GlobalArrayReducer, flow_tools.py:32-47): 24 bytes reach the host, not the grid."""
        expr = self.properties[name]
        f = expr.evaluate() if hasattr(expr, "evaluate") else expr
        ex = self.solver.ex
        if hasattr(f, "require_grid_space") and hasattr(ex, "reduce3"):
            g = f.require_grid_space(f.scales)
            n = int(g.numel()) if hasattr(g, "numel") else int(np.size(g))
            mn, mx, sm = ex.reduce3(g)
            return (mn, mx, sm, n)
        g = np.asarray(f["g"])                       # operands evaluated on the host (analysis-only expressions)
        return (float(g.min()), float(g.max()), float(g.sum()), int(g.size))

    def _sample(self, solver):
        if solver.iteration % self.cadence == 0:
            for name in self.properties:
                self._sampled[name] = self._evaluate(name)

    def _stats(self, name):
        if name in self._sampled:
            return self._sampled[name]
        it = self.solver.iteration
        if self._cache_iter != it:
            self._cache, self._cache_iter = {}, it
        if name not in self._cache:
            self._cache[name] = self._evaluate(name)
        return self._cache[name]

    def _reduce(self, val, op):
        pc = getattr(self.solver.dist, "pcomm", None)
        if pc is None:
            return float(val)
        return pc.allreduce_max(val) if op == "max" else (-pc.allreduce_max(-val) if op == "min" else pc.allreduce_sum(val))

    def min(self, name):
        return self._reduce(self._stats(name)[0], "min")

    def max(self, name):
        return self._reduce(self._stats(name)[1], "max")

    def grid_average(self, name):
        st = self._stats(name)
        return self._reduce(st[2], "sum") / self._reduce(st[3], "sum")

    def volume_integral(self, name):
        from ..core.operators import Integrate
        f = Integrate(self.properties[name]).evaluate()
        return float(np.asarray(f["g"]).ravel()[0])


class CFL:
    """Adaptive timestep from the advective CFL frequency (flow_tools.py:113-233)."""

    def __init__(self, solver, initial_dt, cadence=1, safety=1.0, max_dt=np.inf, min_dt=0.0, max_change=np.inf,
                 min_change=0.0, threshold=0.0):
        self.solver = solver
        self.stored_dt = initial_dt
        self.cadence, self.safety = cadence, safety
        self.max_dt, self.min_dt = max_dt, min_dt
        self.max_change, self.min_change, self.threshold = max_change, min_change, threshold
        self.velocities = []
        self._max_freq = None
        self._setup = {}
        solver._step_hooks.append(self._sample)

    def add_velocity(self, velocity):
        if len(velocity.tensorsig) != 1:
            raise ValueError("Velocity must be a vector")
        self.velocities.append(velocity)

    def add_frequency(self, freq):
        raise NotImplementedError("only velocity-based CFL frequencies are implemented")

    def _spacings(self, u):
        """Device arrays 1/dx per velocity component at the dealias grid (basis.py:6083-6106)."""
        key = id(u)
        if key in self._setup:
            return self._setup[key]
        dist, ex = self.solver.dist, self.solver.ex
        scales = u.domain.dealias
        inv, comp_axis = [], []
        for c in u.tensorsig[0].coords:
            ax = dist.coord_axis(c)
            b = u.domain.by_axis[ax]
            if b is None:
                raise NotImplementedError("CFL along an axis without a basis")
            N = b.grid_size(b.dealias)
            if b.separable:
                dx = np.full(N, b.dealias * b.length / N)
            elif b.a0 == b.b0 == -0.5 and b.a == b.b == -0.5:
                theta = np.pi * (np.arange(N) + 0.5) / N
                dx = b.dealias * b.stretch * np.sin(theta) * np.pi / N
            else:
                dx = np.gradient(b.global_grid(b.dealias), edge_order=2) * b.dealias
            if ax == dist.shard_grid_axis:
                lo, hi = dist.local_block(N)
                dx = dx[lo:hi]
            inv.append(ex.from_host(1.0 / dx))
            comp_axis.append(list(dist.storage_order).index(ax))
        self._setup[key] = (inv, comp_axis, scales)
        return self._setup[key]

    def _sample(self, solver):
        """Called at the start of every step (the reference evaluates its frequency handler there)."""
        if solver.iteration % self.cadence != 0 or not self.velocities:
            return
        ex = solver.ex
        fmax = 0.0
        for u in self.velocities:
            if hasattr(u, "cfl_frequency_max"):             # curvilinear fields carry their own CFL reduction
                fmax = max(fmax, u.cfl_frequency_max())
                continue
            f = u if hasattr(u, "grid_data") else u.evaluate()
            inv, comp_axis, scales = self._spacings(f)
            g = f.grid_data(scales)
            shape = f.domain.storage_grid_shape(scales)
            fmax += ex.cfl_max(g, f.ncomp, shape, inv, comp_axis) if len(self.velocities) == 1 else 0.0
            if len(self.velocities) > 1:
                raise NotImplementedError("several CFL velocities")
        if getattr(solver.dist, "pcomm", None) is not None:
            fmax = solver.dist.pcomm.allreduce_max(fmax)
        self._max_freq = fmax

    def compute_timestep(self):
        """flow_tools.py:183-207"""
        it = self.solver.iteration
        if (it - 1) % self.cadence == 0:
            if (it - 1) <= self.solver.initial_iteration or self._max_freq is None:
                return self.stored_dt
            dt = np.inf if self._max_freq == 0.0 else 1.0 / self._max_freq
            dt *= self.safety
            dt = min(dt, self.max_dt, self.max_change * self.stored_dt)
            dt = max(dt, self.min_dt, self.min_change * self.stored_dt)
            if abs(dt - self.stored_dt) > self.threshold * self.stored_dt:
                self.stored_dt = dt
        return self.stored_dt

    compute_dt = compute_timestep
