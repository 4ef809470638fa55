"""CFL and global flow properties (dedalus/extras/flow_tools.py:49-233), evaluated on the device."""

import numpy as np


class GlobalFlowProperty:
    """Reductions of grid-space expressions every `cadence` iterations (flow_tools.py:49-111)."""

    def __init__(self, solver, cadence=1):
        self.solver = solver
        self.cadence = cadence
        self.properties = {}
        self._cache = {}
        self._cache_iter = None

    def add_property(self, property, name, precompute_integral=False):
        self.properties[name] = property

    def _grid(self, name):
        it = self.solver.iteration
        if self._cache_iter != it:
            self._cache, self._cache_iter = {}, it
        if name not in self._cache:
            expr = self.properties[name]
            f = expr.evaluate() if hasattr(expr, "evaluate") else expr
            f.change_scales(1)
            self._cache[name] = np.array(f["g"])
        return self._cache[name]

    def min(self, name):
        return float(np.min(self._grid(name)))

    def max(self, name):
        return float(np.max(self._grid(name)))

    def grid_average(self, name):
        return float(np.mean(self._grid(name)))

    def volume_integral(self, name):
        from ..core.operators import Integrate
        f = Integrate(self.properties[name]).evaluate()
        return float(np.asarray(f["g"]).ravel()[0])

    def volume_average(self, name):
        from ..core.operators import Average
        f = Average(self.properties[name]).evaluate()
        return float(np.asarray(f["g"]).ravel()[0])


class CFL:
    """Adaptive timestep from the advective CFL frequency (flow_tools.py:113-233;
    AdvectiveCFL core/operators.py:4342-4419, spacings core/basis.py:6078-6113)."""

    def __init__(self, solver, initial_dt, cadence=1, safety=1.0, max_dt=np.inf, min_dt=0.0, max_change=np.inf,
                 min_change=0.0, threshold=0.0):
        self.solver = solver
        self.stored_dt = initial_dt
        self.cadence, self.safety = cadence, safety
        self.max_dt, self.min_dt = max_dt, min_dt
        self.max_change, self.min_change, self.threshold = max_change, min_change, threshold
        self.velocities = []

    def add_velocity(self, velocity):
        self.velocities.append(velocity)

    def add_frequency(self, freq):
        raise NotImplementedError

    def _max_frequency(self):
        dist = self.solver.dist
        fmax = 0.0
        for u in self.velocities:
            f = u.evaluate() if not hasattr(u, "fill_random") else u
            f.change_scales(1)
            g = np.asarray(f["g"])
            freq = np.zeros(g.shape[1:])
            cs = f.tensorsig[0]
            for i, c in enumerate(cs.coords):
                ax = dist.coord_axis(c)
                b = f.domain.by_axis[ax]
                if b is None:
                    continue
                if b.separable:
                    # Fourier: dx = L / N_grid at scale 1 (uniform)
                    spacing = np.full(b.grid_size(1), b.length / b.grid_size(1))
                else:
                    spacing = np.gradient(b.global_grid(1), edge_order=2)
                shape = [1] * dist.dim
                shape[ax] = spacing.size
                freq = freq + np.abs(g[i]) / spacing.reshape(shape)
            fmax = max(fmax, float(freq.max()))
        return fmax

    def compute_timestep(self):
        """flow_tools.py:183-220"""
        it = self.solver.iteration
        if (it - self.solver.initial_iteration) % self.cadence == 0:
            fmax = self._max_frequency()
            dt = self.safety / fmax if fmax > 0 else np.inf
            dt = min(dt, self.max_dt)
            dt = max(dt, self.min_dt)
            if self.stored_dt is not None and np.isfinite(self.stored_dt):
                dt = min(dt, self.max_change * self.stored_dt)
                dt = max(dt, self.min_change * self.stored_dt)
                if abs(dt - self.stored_dt) / self.stored_dt < self.threshold:
                    dt = self.stored_dt
            self.stored_dt = dt
        return self.stored_dt
