"""
Life cycle shared by the three initial-value solvers (Cartesian pencils, sphere, shell): the reference has ONE
InitialValueSolver (core/solvers.py:503-806) and its stop conditions, clocks, step bookkeeping, Hermitian-symmetry
schedule, evolve loop and statistics apply to every geometry alike.

Clocks are WORLD clocks: on several ranks rank 0's time is broadcast (core/solvers.py:603-611), so every rank takes
the same stop / wall_dt-schedule decision and the collectives inside a step (transposes, gathered output) always
match up.
"""

import logging
import time

import numpy as np

logger = logging.getLogger("solvers")


class IVPLifecycle:
    enforce_real_cadence = 100
    warmup_iterations = 10

    def _init_lifecycle(self, enforce_real_cadence=100, warmup_iterations=10):
        self.iteration = self.initial_iteration = 0
        self.stop_sim_time = np.inf
        self.stop_wall_time = np.inf
        self.stop_iteration = np.inf
        self.enforce_real_cadence = enforce_real_cadence
        self.warmup_iterations = warmup_iterations
        self.dt = None
        self.init_time = self.start_time = self.world_time          # (start_time: older name, kept for scripts)
        self.warmup_time = None
        self.run_time_start = None

    # ---- clocks ------------------------------------------------------------------------------------------------
    @property
    def world_time(self):
        """core/solvers.py:603-611: root's clock on every rank"""
        t = time.time()
        pcomm = getattr(self.dist, "pcomm", None)
        if pcomm is not None and getattr(self.dist, "size", 1) > 1 and self._clock_decides():
            t = pcomm.bcast_float(t)
        return t

    def _clock_decides(self):
        """Does any decision depend on the wall clock (stop_wall_time, a handler scheduled by wall_dt)?  Only then is the
        broadcast worth its price: on the GPUs it synchronises host and device once per step."""
        if np.isfinite(getattr(self, "stop_wall_time", np.inf)):
            return True
        ev = getattr(self, "evaluator", None)
        return any(getattr(h, "wall_dt", None) for h in getattr(ev, "handlers", ()))

    @property
    def wall_time(self):
        """Seconds elapsed since instantiation (core/solvers.py:613-616)."""
        return self.world_time - self.init_time

    # ---- stop conditions -----------------------------------------------------------------------------------------
    @property
    def proceed(self):
        """core/solvers.py:618-630"""
        if self.sim_time >= self.stop_sim_time:
            logger.info("Simulation stop time reached.")
            return self._stopped()
        if self.wall_time >= self.stop_wall_time:
            logger.info("Wall stop time reached.")
            return self._stopped()
        if self.iteration >= self.stop_iteration:
            logger.info("Stop iteration reached.")
            return self._stopped()
        return True

    def _stopped(self):
        self._flush_outputs()            # staged analysis output reaches its files when the main loop ends
        return False

    def _flush_outputs(self):
        ev = getattr(self, "evaluator", None)
        if ev is not None and hasattr(ev, "flush"):
            ev.flush()

    # ---- one step --------------------------------------------------------------------------------------------------
    def step(self, dt):
        """Advance one timestep (core/solvers.py:683-711)."""
        if not np.isfinite(dt):
            raise ValueError("Invalid timestep")
        wall_time = self.wall_time
        if self.iteration == self.initial_iteration + self.warmup_iterations:
            self.ex.sync()
            self.warmup_time = self.world_time
            self.run_time_start = wall_time
        if self.dt is None:
            self.dt = dt
        # scheduled analysis sees the pre-step state (the reference's timesteppers call evaluate_scheduled first,
        # core/timesteppers.py:137-139, 578-580), with the same world wall time on every rank
        self._step_wall_time = wall_time
        self._step_dt = dt
        for hook in self._step_hooks:
            hook(self)
        if not self._graph_replay(dt, wall_time):
            self.timestepper.step(dt, wall_time)
        # Hermitian symmetry for real variables: for as many iterations as the scheme uses internally, every cadence
        # (core/solvers.py:704-708) -- checked BEFORE the iteration count is advanced, so the first step takes part
        if self.enforce_real_cadence and self.iteration % self.enforce_real_cadence < self.timestepper.steps:
            self.enforce_hermitian_symmetry(self.state)
        self.iteration += 1
        self.dt = dt

    # ---- fixed-timestep steps as one HIP graph ------------------------------------------------------------------------
    # Small problems are launch bound (2-D Rayleigh-Benard 512 x 256: ~75 % of a step is host time issuing ~100
    # launches).  With a fixed timestep a Runge-Kutta step is the same launch sequence on the same buffers every time:
    # it is captured once (torch.cuda.CUDAGraph = hipGraph; every kernel of libdedalus_hip runs on the capturing
    # stream) and replayed.  Anything that breaks the pattern -- a different dt, state touched through the host or the
    # grid, several ranks, a scheme that rotates its history buffers -- falls back to normal launches.
    step_graph = None            # None: decide from DDH_STEP_GRAPH (default off); True / False: forced

    def enable_step_graph(self, on=True):
        self.step_graph = bool(on)
        self._graph = None

    # Solvers whose steps are launch bound replay them by default (Cartesian problems of up to 2^22 modes: KdV-Burgers N =
    # 1024 runs 3.4 x faster, 2-D Rayleigh-Benard 512 x 256 is ~100 launches of a few microseconds); DDH_STEP_GRAPH=1 / 0
    # or enable_step_graph() force it either way.  Larger problems gain nothing (the host already runs ahead of the device).
    step_graph_auto_modes = 0          # > 0: replay by default when total_modes <= this (set by the Cartesian IVP solver)

    def _graph_wanted(self):
        if self.step_graph is None:
            import os
            env = os.environ.get("DDH_STEP_GRAPH")
            if env is not None:
                self.step_graph = env == "1"
            else:
                self.step_graph = 0 < getattr(self, "total_modes", 0) <= self.step_graph_auto_modes
        return self.step_graph

    def _state_is_clean(self):
        for f in self.state:
            if getattr(f, "_authority", "device") != "device" or getattr(f, "layout", "c") != "c":
                return False
            if getattr(f, "_host_dirty", False):         # (a constant the user has looked at or set: uploaded by an ordinary step)
                return False
        return True

    def _graph_replay(self, dt, wall_time):
        """-> True when the step was advanced by replaying (or capturing + replaying) a graph"""
        if not self._graph_wanted() or getattr(self.ex, "name", "") != "hip" or getattr(self.dist, "size", 1) > 1:
            return False
        from .timesteppers import RungeKuttaIMEX, MultistepIMEX
        ts = self.timestepper
        if not isinstance(ts, (RungeKuttaIMEX, MultistepIMEX)) or getattr(self.ex, "timer", None) is not None:
            return False
        st = getattr(self, "_graph", None)
        if st is None:
            st = self._graph = dict(dt=None, seen=0, graphs={}, failed=False)
        if st["failed"] or not self._state_is_clean() or getattr(self, "solve_probe", None) is not None:
            return False
        if dt != st["dt"]:                   # (re)start: two ordinary steps with this dt first (factorization, plans, buffers)
            st.update(dt=dt, seen=0, graphs={})
        # Runge-Kutta: one graph.  Multistep: the history buffers rotate, one graph per phase of the rotation -- and only
        # once the start-up orders are over and the whole timestep history equals dt (the coefficients are constant then)
        phase = (1, 0)
        if isinstance(ts, MultistepIMEX):
            phase = ts.graph_phase(dt)
            if phase is None:
                return False
        g = st["graphs"].get(phase[1])
        if g is None:
            st["seen"] += 1
            if st["seen"] <= 2:
                return False
            torch = self.ex.torch
            t0 = self.sim_time
            snap = ts.graph_snapshot() if hasattr(ts, "graph_snapshot") else None
            # Cyclic garbage must not be collected INSIDE the capture: a finalizer that releases device memory (an old
            # solver's plans, factorizations: ddh_destroy -> hipFree) is illegal on a capturing stream and aborts the process.
            import gc
            gc_was_on = gc.isenabled()
            try:
                g = torch.cuda.CUDAGraph()
                gc.collect()
                torch.cuda.synchronize()
                gc.disable()
                try:
                    with torch.cuda.graph(g):
                        ts.step(dt, wall_time)
                finally:
                    if gc_was_on:
                        gc.enable()
                st["graphs"][phase[1]] = g
            except Exception as e:                      # something in the step is not capturable: never try again
                st["failed"] = True
                logger.warning("step graph capture failed (%s): falling back to ordinary launches" % (e,))
                self.sim_time = t0
                if snap is not None:                    # the aborted capture rotated the history without running a kernel
                    ts.graph_rollback(snap)
                return False
            self.sim_time = t0                          # capture only recorded the launches (the host side of the step ran)
            g.replay()
            self.sim_time = t0 + dt
            return True
        t0 = self.sim_time
        g.replay()
        if isinstance(ts, MultistepIMEX):
            ts.graph_advance(dt)
        self.sim_time = t0 + dt
        return True

    def enforce_hermitian_symmetry(self, fields):
        """Grid and back at the dealias scales (core/solvers.py:675-681)."""
        for f in fields:
            self._hermitian_round_trip(f)

    def _hermitian_round_trip(self, f):
        raise NotImplementedError

    # ---- main loop ---------------------------------------------------------------------------------------------------
    def evolve(self, timestep_function, log_cadence=100):
        """core/solvers.py:713-735"""
        if np.isinf(self.stop_sim_time) and np.isinf(self.stop_wall_time) and np.isinf(self.stop_iteration):
            raise ValueError("No stopping criterion specified.")
        try:
            logger.info("Starting main loop")
            while self.proceed:
                timestep = timestep_function()
                self.step(timestep)
                if (self.iteration - 1) % log_cadence == 0:
                    logger.info("Iteration=%i, Time=%e, Step=%e" % (self.iteration, self.sim_time, timestep))
        except Exception:
            logger.error("Exception raised, triggering end of main loop.")
            raise
        finally:
            self._flush_outputs()
            self.log_stats()

    def log_stats(self, format=".4g"):
        """core/solvers.py:755-778: setup / warm-up / run time and mode-stages per (device-)second."""
        self.ex.sync()
        self._flush_outputs()
        end = self.world_time
        logger.info("Final iteration: %i" % self.iteration)
        logger.info("Final sim time: %s" % self.sim_time)
        logger.info("Setup time (init - iter 0): %.4g sec" % getattr(self, "setup_time", 0.0))
        if self.warmup_time is not None:
            run = end - self.warmup_time
            its = self.iteration - self.initial_iteration - self.warmup_iterations
            stages = its * self.timestepper.stages
            logger.info("Run time (iter %d-end): %.4g sec" % (self.warmup_iterations, run))
            if run > 0:
                logger.info("Speed: %.4g mode-stages/gpu-sec" % (self.total_modes * stages / run))
        else:
            logger.info("Timings unavailable because warmup did not complete.")

    def load_state(self, path, index=-1, allow_missing=False):
        """core/solvers.py:632-673"""
        from .output import load_state
        return load_state(self, path, index=index, allow_missing=allow_missing)
