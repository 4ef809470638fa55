"""
Distributor: owns the coordinate system, the storage-order convention, the device executor and the
per-axis transform driver.  Replaces the Layout / Transform / Transpose path machinery of
dedalus/core/distributor.py:36-175, 311-517, 588-661 (one rank per GPU; the pencil transposes of
:696-924 become RCCL all-to-alls, see parallel.py).
"""

import os

import numpy as np

from .coords import CartesianCoordinates, Coordinate
from .field import Field


class _RankInfo:
    """What scripts read from dist.comm / dist.comm_cart when no MPI communicator is passed (one process per GPU;
    the collectives themselves go through torch.distributed, parallel.py)."""

    def __init__(self, rank, size):
        self.rank, self.size = rank, size

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size


class Distributor:
    def __init__(self, coordsystems, comm=None, mesh=None, dtype=None, executor=None):
        if isinstance(coordsystems, (CartesianCoordinates, Coordinate)):
            coordsystems = (coordsystems,)
        self.coordsystems = tuple(coordsystems)
        self.coords = tuple(c for cs in self.coordsystems for c in cs.coords)
        self.dim = len(self.coords)
        self.dtype = np.dtype(np.float64 if dtype is None else dtype)
        self.mesh = tuple(int(m) for m in mesh) if mesh is not None else ()
        self.size = int(np.prod(self.mesh)) if self.mesh else 1
        self.rank = 0
        if self.size > 1:
            if len(self.mesh) != 1:
                raise NotImplementedError("only 1-D process meshes are implemented in this round")
            from ..parallel import Comm
            self.pcomm = Comm(self.size)
            self.rank = self.pcomm.rank
        else:
            self.pcomm = None
        self._jacobi_axes = set()
        self._layout_frozen = False
        self._executor = executor
        self.transformer = Transformer(self)
        self.comm = comm if comm is not None else _RankInfo(self.rank, self.size)

    # ---- executor (device) -------------------------------------------------------------------------
    @property
    def executor(self):
        if self._executor is None:
            from ..executor import HipExecutor
            self._executor = HipExecutor()       # raises without a gfx950 device: no CPU fallback
        return self._executor

    # ---- coordinates ---------------------------------------------------------------------------------
    def coord_axis(self, coord):
        return self.coords.index(coord)

    def get_coord(self, name):
        for c in self.coords:
            if c.name == name:
                return c
        raise ValueError("unknown coordinate %r" % name)

    def get_axis(self, coord):
        return self.coord_axis(coord)

    # ---- storage order ---------------------------------------------------------------------------------
    def _register_domain(self, domain):
        for ax, b in enumerate(domain.by_axis):
            if b is not None and not b.separable and ax not in self._jacobi_axes:
                if self._layout_frozen:
                    raise ValueError("a Jacobi basis appeared on a new axis after fields were allocated")
                self._jacobi_axes.add(ax)
        if sum(b is not None for b in domain.by_axis) > 1:
            self._layout_frozen = True
        if len(self._jacobi_axes) > 1:
            raise NotImplementedError("at most one Jacobi (coupled) axis is supported in this round")

    @property
    def storage_order(self):
        jac = sorted(self._jacobi_axes)
        return tuple(jac + [ax for ax in range(self.dim) if ax not in self._jacobi_axes])

    @property
    def separable_axes(self):
        """User axes of the separable (Fourier) directions, in storage order."""
        return tuple(ax for ax in self.storage_order if ax not in self._jacobi_axes)

    # ---- sharding (1-D mesh): coefficient space along the first separable axis, grid space along the
    #      Jacobi axis; everything else is local
    @property
    def shard_coeff_axis(self):
        sep = self.separable_axes
        return sep[0] if (self.size > 1 and sep) else None

    @property
    def shard_grid_axis(self):
        jac = sorted(self._jacobi_axes)
        return jac[0] if (self.size > 1 and jac) else None

    def local_block(self, n):
        """(start, stop) of this rank's block of an axis of length n."""
        if n % self.size:
            raise ValueError("axis of length %d is not divisible by the %d ranks" % (n, self.size))
        blk = n // self.size
        return self.rank * blk, (self.rank + 1) * blk

    def coupled_size(self, domain):
        for ax in self._jacobi_axes:
            b = domain.by_axis[ax]
            if b is not None:
                return b.coeff_size
        return 1

    # ---- field factories (core/distributor.py:Field/VectorField/TensorField) ----------------------------
    def Field(self, *args, **kw):
        return Field(self, *args, **kw)

    ScalarField = Field

    def VectorField(self, coordsys, *args, **kw):
        return Field(self, *args, tensorsig=(coordsys,), **kw)

    def TensorField(self, coordsys, *args, order=2, **kw):
        if isinstance(coordsys, (tuple, list)):
            sig = tuple(coordsys)
        else:
            sig = (coordsys,) * order
        return Field(self, *args, tensorsig=sig, **kw)

    # ---- grids ---------------------------------------------------------------------------------------
    def _reshape_axis(self, arr, axis):
        shape = [1] * self.dim
        shape[axis] = arr.size
        return arr.reshape(shape)

    def local_grid(self, basis, scale=None):
        scale = 1.0 if scale is None else scale
        return self.local_grids(basis, scales=scale)[0]

    def local_grids(self, *bases, scales=None):
        out = []
        for b in bases:
            ax = self.coord_axis(b.coord)
            if scales is None:
                s = 1.0
            elif isinstance(scales, (tuple, list)):
                s = scales[ax]
            else:
                s = scales
            g = b.global_grid(s)
            if ax == self.shard_grid_axis:
                lo, hi = self.local_block(g.size)
                g = g[lo:hi]
            out.append(self._reshape_axis(g, ax))
        return tuple(out)

    def local_modes(self, basis):
        ax = self.coord_axis(basis.coord)
        if basis.separable:
            return self._reshape_axis(basis.wavenumbers, ax)
        return self._reshape_axis(np.arange(basis.size), ax)


def _overlap():
    """Exchange the components of a field one by one with the packing / unpacking of its neighbours overlapped (the
    exchange of component c is in flight on the communicator's stream while c + 1 is packed and c - 1 unpacked).
    On by default since round 3 (the 2-rank parity tests run with it, tests/test_multiprocess.py); DDH_A2A_OVERLAP=0
    sends a whole field in one exchange through the library's RCCL plan (ddh_a2a_localize_*)."""
    import os
    return os.environ.get("DDH_A2A_OVERLAP", "1") == "1"


class Transformer:
    """Walks a field through its axes (the reference's Transform.increment/decrement loop,
    core/distributor.py:601-661): backward = Jacobi axis first then Fourier axes in storage order,
    forward = the reverse.  Every step is one out-of-place kernel launch."""

    def __init__(self, dist):
        self.dist = dist

    def _steps(self, domain, scales):
        """[(storage position, basis, plan_spec)] for axes that carry a basis (cached: asked dozens of times per step)."""
        cache = self.__dict__.setdefault("_steps_cache", {})
        key = (id(domain), tuple(scales))
        hit = cache.get(key)
        if hit is not None and hit[0] is domain:
            return hit[1]
        steps = []
        for pos, ax in enumerate(self.dist.storage_order):
            b = domain.by_axis[ax]
            if b is not None:
                steps.append((pos, b, b.plan_spec(scales[ax])))
        if len(cache) > 256:
            cache.clear()
        cache[key] = (domain, steps)
        return steps

    def backward(self, field, c, g, scales):
        return self.backward_data(field.domain, field.ncomp, c, g, scales)

    def forward(self, field, g, scales, c):
        return self.forward_data(field.domain, field.ncomp, g, scales, c)

    def _needs_exchange(self, domain):
        d = self.dist
        if d.size == 1:
            return False
        a, z = d.shard_coeff_axis, d.shard_grid_axis
        has_x = a is not None and domain.by_axis[a] is not None
        has_z = z is not None and domain.by_axis[z] is not None
        if has_x and has_z:
            return True
        if has_x != has_z:
            raise NotImplementedError("grid-space data of a field that lacks the x or z basis on several ranks")
        return False

    def fusable_last_axis(self, domain, scales):
        """plan spec of the last storage axis when the grid stage can be fused along it (a contiguous
        RealFourier axis), else None."""
        steps = self._steps(domain, scales)
        if not steps:
            return None
        pos, b, spec = steps[-1]
        if pos != len(self.dist.storage_order) - 1 or spec[0] != "rfft":
            return None
        if not self.dist.executor.fused_capable(spec):
            return None
        return b, spec

    def pregrid_shape(self, domain, ncomp, scales, window=True):
        """[comp][grid...][last axis coefficients]: every axis but the last in grid space (local shape; window: of the
        window of z planes being evaluated, when there is one)."""
        shape = [ncomp] + list(domain.storage_grid_shape(scales))
        b = domain.by_axis[self.dist.storage_order[-1]]
        if b is not None:
            shape[-1] = b.coeff_size            # full size: this layout sits on the grid side of the exchange
        win = self.__dict__.get("_win")
        if window and win is not None and win["k"] is not None and len(shape) == 4:
            shape[1] = shape[1] // win["K"]     # a window of this rank's z planes (windows, below)
        return tuple(shape)

    # ---- the grid stage in windows of z planes (several ranks, blocked stage layout) -------------------------------------
    # Everything between the two pencil transposes -- x backward, fused y stage, x forward: two thirds of a rank's kernels --
    # is independent from z plane to z plane, and a window of planes of the exchanged layout [p][z_loc][B][ky] is one
    # contiguous range per peer.  Solver.evaluate_F therefore runs that stage K times on 1 / K of the planes: the z steps of
    # every field first (their exchanges are REGISTERED here, not started), then all parts of window 0, 1, .. queued on the
    # communicator's stream (ddh_comm_alltoall_part), then per window the x steps -- each waits for its own parts only --,
    # the fused launch, and the x forward step, whose window is on the wire while the next window computes.  Exposed wire:
    # the first window's arrivals and the last window's departures instead of everything (profiles/r6_rank_emulation.txt).
    def windows_begin(self, K):
        self._win = dict(K=int(K), k=None, pending=[], works={}, arrays={})

    def windows_end(self):
        win = self.__dict__.get("_win")
        if win is not None:
            for w in win["works"].values():
                w.wait()
        self._win = None

    def windows_start_exchanges(self):
        """Queue the registered backward exchanges window by window (all components of all fields of window 0 first)."""
        win = self._win
        P = self.dist.size
        pc = self.dist.pcomm
        for k in range(win["K"]):
            for src, out2, nc, gz, per_plane in win["pending"]:
                cw = gz // win["K"]
                # the components of a field as ONE group of sends / receives
                win["works"][(out2.data_ptr(), k)] = pc.all_to_all_start(
                    out2.reshape(-1), src.reshape(-1), part=(k * cw * per_plane, cw * per_plane), batch=nc)
                pc.stats["exchanges"] += nc
        win["pending"] = []

    def window_set(self, k):
        self._win["k"] = k

    def _window_of(self, src):
        """(k, K, works) when src is a stage-1 array exchanged in windows and a window is being evaluated."""
        win = self.__dict__.get("_win")
        if win is None or win["k"] is None or src.data_ptr() not in win["arrays"]:
            return None
        return win

    def nsteps(self, domain, scales):
        return len(self._steps(domain, scales))

    def stage_shape(self, domain, ncomp, scales, k):
        """Local shape after the first k backward steps (k = 0: coefficient space)."""
        steps = self._steps(domain, scales)
        shape = [ncomp] + list(domain.storage_coeff_shape())
        exchange = self._needs_exchange(domain)
        for (pos, b, spec) in steps[:k]:
            ax = self.dist.storage_order[pos]
            shape[pos + 1] = b.grid_size(scales[ax])
            if exchange and pos == 0:
                P = self.dist.size
                shape[1], shape[2] = shape[1] // P, shape[2] * P
        return tuple(shape)

    def stage_xb(self, domain, scales):
        """(z-side value, x-side value) when the array between the z and the x transforms of `domain` at `scales` -- "stage
        1": after backward step 0, before forward step 0 -- is stored blocked, else None.  A RULE, not a property of an
        array: every producer and consumer of a stage-1 array asks it with the same arguments.
          * one rank: [comp][kx / 64][z][kx % 64][ky] (include/dedalus_hip.h, ddh_fft_set_stage_layout) -> (ny, gz): the z
            transforms write / read it on their grid side, the x transforms on their coefficient side;
          * several ranks (round 6): the all-to-all delivers a component as [p][z_loc][nx / P][ky]; with blocks of nx / P rows
            that IS the blocked layout on the x side -> (0, (gz_loc, nx / P)): the z side stays natural (it is what is sent
            as it lies), the x transforms read what arrived and write what leaves, and the unpack / pack passes around the
            exchange disappear (ddh_fft_set_stage_block).  DDH_A2A_BLOCKED=0 keeps the natural layout + unpack / pack.
        A domain with bases on all three axes, both transforms on the strided wave kernels.  DDH_NO_STAGE_XB=1 switches the
        one-rank layout off."""
        cache = self.__dict__.setdefault("_xb_cache", {})
        key = (id(domain), tuple(scales))
        hit = cache.get(key)
        if hit is not None and hit[0] is domain:
            return hit[1]
        res = None
        ex = self.dist.executor
        steps = self._steps(domain, scales)
        if len(steps) == 3 and hasattr(ex, "stage_layout_ok") and [p for p, _, _ in steps] == [0, 1, 2]:
            (_, bz, zspec), (_, bx, xspec), (_, by, _) = steps
            shape = domain.storage_coeff_shape()
            nx, ny = int(shape[1]), int(shape[2])
            gz = int(bz.grid_size(scales[self.dist.storage_order[0]]))
            P = self.dist.size
            if P == 1:
                if os.environ.get("DDH_NO_STAGE_XB") is None and ex.stage_layout_ok(zspec, xspec, nx, ny):
                    res = (ny, gz)
            elif (os.environ.get("DDH_A2A_BLOCKED", "1") != "0" and _overlap() and self._needs_exchange(domain)
                  and hasattr(ex, "stage_block_ok") and gz % P == 0 and ex.stage_block_ok(xspec, nx * P, nx, ny)):
                res = (0, (gz // P, nx))                 # (nx: this rank's rows = nx_global / P)
        cache[key] = (domain, res)
        return res

    # ---- exchanges in flight (blocked stage layout, several ranks) ------------------------------------------------------
    # The z step starts one exchange per component on the communicator's stream and returns; the first x-side consumer of
    # a component waits for it (a stream-side wait: the host never blocks).  With the evaluator issuing every field's z
    # step first (Evaluator.prefetch_stage1) the wire of a later field runs under the x transforms of an earlier one.
    def _defer(self, arr, works):
        self.__dict__.setdefault("_inflight", []).append([arr.data_ptr(), arr[0:1].numel() * 8, list(works), arr])

    def in_flight(self, arr):
        """True when an exchange into (part of) arr has not been waited for."""
        fl = self.__dict__.get("_inflight")
        if not fl:
            return False
        lo = arr.data_ptr()
        hi = lo + arr.numel() * 8
        return any(w is not None and base + c * cb < hi and lo < base + (c + 1) * cb
                   for base, cb, works, _ in fl for c, w in enumerate(works))

    def wait_for(self, arr):
        """Make the current stream wait for the exchanges that deliver (part of) arr."""
        fl = self.__dict__.get("_inflight")
        if not fl:
            return
        lo = arr.data_ptr()
        hi = lo + arr.numel() * 8
        keep = []
        for ent in fl:
            base, cb, works, _ = ent
            for c, w in enumerate(works):
                if w is not None and base + c * cb < hi and lo < base + (c + 1) * cb:
                    w.wait()
                    works[c] = None
            if any(w is not None for w in works):
                keep.append(ent)
        self._inflight = keep

    def settle(self):
        """Wait for every exchange still in flight (end of an evaluation pass)."""
        for _, _, works, _ in self.__dict__.get("_inflight", ()):
            for w in works:
                if w is not None:
                    w.wait()
        self._inflight = []

    def _x_step_by_component(self, src, ncomp):
        """DDH_A2A_SPLIT_X=1: the x step runs component by component when its input is still arriving (component c is
        transformed while component c + 1 is on the wire).  Off by default: under an emulated 75 GB/s wire the smaller
        launches cost what the overlap wins (P = 4: 30.1 ms per step without, 31.0 with; profiles/r6_rank_emulation.txt)."""
        return ncomp > 1 and os.environ.get("DDH_A2A_SPLIT_X", "0") == "1" and self.in_flight(src)

    def coeff_tiled_ok(self, domain, scales, row_len):
        """Can the first backward step of `domain` read coefficient rows [nx][row_len] stored tile-major?  (a strided
        Chebyshev step on the wave kernels, one rank: Executor.tiled_forward_ok states the sizes)"""
        ex = self.dist.executor
        steps = self._steps(domain, scales)
        if self.dist.size > 1 or not steps or not hasattr(ex, "tiled_forward_ok") or not hasattr(ex, "tile_rows"):
            return False
        pos, b, spec = steps[0]
        shape = domain.storage_coeff_shape()
        inner = int(np.prod(shape[pos + 1:]))
        return pos == 0 and len(shape) == 3 and int(shape[2]) == row_len and ex.tiled_forward_ok(spec, b, inner, row_len)

    def backward_steps(self, domain, ncomp, src, scales, i0, i1, dst=None, deriv=None, ctile=0):
        """Apply backward steps i0 .. i1-1 to data that has seen steps < i0 (z transform first, then the
        all-to-all (-> z-sharded, kx local), then the Fourier transforms).  deriv = (step, 2 pi / L)
        differentiates along that RealFourier step's axis while its coefficients are loaded.  ctile (with i0 = 0): the
        coefficient rows of src are tile-major, rows of that length (coeff_tiled_ok).
        Returns the result (dst when given)."""
        ex = self.dist.executor
        steps = self._steps(domain, scales)
        shape = list(self.stage_shape(domain, ncomp, scales, i0))
        exchange = self._needs_exchange(domain)
        for i in range(i0, i1):
            pos, b, spec = steps[i]
            ax = self.dist.storage_order[pos]
            n_out = b.grid_size(scales[ax])
            outer = int(np.prod(shape[:pos + 1]))
            inner = int(np.prod(shape[pos + 2:]))
            shape[pos + 1] = n_out
            last = (i == i1 - 1)
            out = dst if (last and dst is not None and not (exchange and pos == 0)) else ex.empty(tuple(shape))
            xb = self.stage_xb(domain, scales) if i <= 1 else None
            xbv = dict(xb=xb[i]) if (xb is not None and xb[i]) else {}      # step 0 writes, step 1 reads the stage-1 array
            if deriv is not None and deriv[0] == i:
                if spec[0] != "rfft":
                    raise NotImplementedError("derivative at load along a non-Fourier axis")
                xbv["deriv"] = deriv[1]
            if ctile and i == 0:
                xbv["tiled_row"] = ctile
            win = self._window_of(src) if (exchange and pos == 1) else None
            if win is not None:
                # the x step on ONE window of this rank's planes: component by component (a window of a component is what
                # a launch can address), each waiting for its own part of the exchange; the result has cw planes
                _, gz, rows, rest = win["arrays"][src.data_ptr()]
                cw = gz // win["K"]
                z0 = win["k"] * cw
                shape[1] = cw
                out = ex.empty(tuple(shape))
                w = win["works"].pop((src.data_ptr(), win["k"]), None)
                if w is not None:
                    w.wait()
                xbv["xb"] = tuple(xbv["xb"]) + (z0, cw)
                ex.transform(spec, b, "backward", src, out, ncomp * cw, inner, **xbv)
            elif exchange and pos == 1 and self.__dict__.get("_win") is not None and self._win["k"] is not None:
                raise RuntimeError("windowed evaluation: an x step on a stage array that was not exchanged in windows")
            elif exchange and i >= 1 and self._x_step_by_component(src, ncomp):
                for c in range(ncomp):
                    self.wait_for(src[c:c + 1])
                    ex.transform(spec, b, "backward", src[c:c + 1], out[c:c + 1], outer // ncomp, inner, **xbv)
            else:
                if exchange and i >= 1:
                    self.wait_for(src)
                ex.transform(spec, b, "backward", src, out, outer, inner, **xbv)
            src = out
            if exchange and pos == 0:
                src = self._rows_after_z(ex, src, shape, dst if last else None,
                                         blocked=self.stage_xb(domain, scales) is not None)
        return src

    def _exchange(self, ex, which, src, dst, n0, n1, n2, n3):
        """One pencil transpose of the global [n0][n1][n2][n3] view (n1 = Jacobi axis, n2 = first Fourier axis):
        "rows":    CL [n0][n1][n2/P][n3] -> RL [n0][n1/P][n2][n3]   (backward direction here: kx-sharded -> z-sharded)
        "columns": RL -> CL                                          (forward direction)
        Through the library's RCCL plan (ddh_a2a_localize_*) on the GPUs; through torch.distributed around the
        ddh_a2a_pack / unpack kernels in the gloo test configurations and on the CPU oracle."""
        P = self.dist.size
        pcomm = self.dist.pcomm
        nbytes = 8 * int(n0 * n1 * n2 * n3) // P
        timer = getattr(ex, "timer", None)

        def run():
            plan = ex.a2a_plan(pcomm, n0, n1, n2, n3) if hasattr(ex, "a2a_plan") else None
            if plan is not None:
                (ex.a2a_localize_rows if which == "rows" else ex.a2a_localize_columns)(plan, src, dst)
                pcomm.note_via("ddh_a2a_localize (%s plan)" % ("loop-back" if pcomm.backend == "loopback" else "library RCCL"),
                               nbytes * (P - 1) // P)
                return
            send = ex.empty((nbytes // 8,))
            recv = ex.empty((nbytes // 8,))
            if which == "rows":
                ex.a2a_pack(src, send, n0, n1, n2 // P, n3, P)
                pcomm.all_to_all(recv, send)
                ex.a2a_unpack(recv, dst, n0, n1 // P, n2, n3, P)
            else:
                ex.a2a_pack(src, send, n0 * (n1 // P), n2, 1, n3, P)
                pcomm.all_to_all(recv, send)
                ex.a2a_unpack(recv, dst, n0, 1, n1, (n2 // P) * n3, P)

        pcomm.stats["exchanges"] += 1
        pcomm.stats["bytes_sent"] += nbytes * (P - 1) // P
        if timer is not None:
            timer.run("a2a_exchange", 2 * nbytes, run)         # pack read+write, unpack read+write ~ 4x; wire 1x
        else:
            run()

    def _rows_after_z(self, ex, src, shape, dst=None, blocked=False):
        """The pencil transpose that follows the backward z transform: [comp, Gz, nx_loc, ny] -> [comp, Gz/P, nx, ny]
        (`shape` is updated in place)."""
        P = self.dist.size
        nc, Gz, nxl = shape[0], shape[1], shape[2]
        rest = int(np.prod(shape[3:]))
        shape[1], shape[2] = Gz // P, nxl * P
        out2 = dst if dst is not None else ex.empty(tuple(shape))
        if blocked:
            # blocked x side (stage_xb, several ranks): a component is sent as it lies and received as [p][Gz / P][nx_loc][..]
            # straight into its slice of the result -- which the x transforms read in that layout: no kernel at all
            win = self.__dict__.get("_win")
            if win is not None and win["k"] is None and dst is None and (Gz // P) % win["K"] == 0:
                # windows: registered, started window by window once every field's z step is issued
                win["pending"].append((src, out2, nc, Gz // P, nxl * rest))
                win["arrays"][out2.data_ptr()] = (out2, Gz // P, nxl, rest)
                return out2
            # (the components of a field as ONE group of sends / receives: ddh_comm_alltoall_part with the whole block as part)
            w = self.dist.pcomm.all_to_all_start(out2.reshape(-1), src.reshape(-1), part=(0, (Gz // P) * nxl * rest), batch=nc)
            works = [w] * nc
            self.dist.pcomm.stats["exchanges"] += nc
            if os.environ.get("DDH_A2A_DEFER", "1") != "0" and hasattr(out2, "data_ptr"):
                self._defer(out2, works)                 # waited for by the x step that reads it (wait_for)
            else:
                for w in works:
                    w.wait()
            return out2
        if nc > 1 and _overlap():
            # per-component pipeline: the exchange of component c runs (on the communicator's stream) while
            # component c + 1 is packed and component c - 1 is unpacked
            n1 = int(Gz * nxl * rest)
            pend = None
            for c in range(nc):
                # one component [Gz][nx_loc][rest] split along z into P blocks IS the packed order [p][Gz / P][nx_loc][rest]:
                # it is sent as it lies (round 5 ran a pack kernel here that copied the component onto itself)
                send = src[c:c + 1].reshape(-1)
                recv = ex.empty((n1,))
                work = self.dist.pcomm.all_to_all_start(recv, send)
                if pend is not None:
                    pend[0].wait()
                    ex.a2a_unpack(pend[1], out2[pend[2]:pend[2] + 1], 1, Gz // P, nxl * P, rest, P)
                pend = (work, recv, c, send)
            pend[0].wait()
            ex.a2a_unpack(pend[1], out2[pend[2]:pend[2] + 1], 1, Gz // P, nxl * P, rest, P)
        else:
            self._exchange(ex, "rows", src, out2, nc, Gz, nxl * P, rest)
        return out2

    def backward_dual_z(self, ldomain, xdomain, ncomp, src, scales, dvec, ctile=0):
        """The backward Jacobi-axis step (step 0) of a field's coefficients twice from one read: -> (the field's transform,
        the transform of the one-superdiagonal operator `dvec` (device array) applied to the coefficients, taken in
        xdomain's basis), each followed by the pencil transpose on several ranks."""
        ex = self.dist.executor
        pos, b, spec = self._steps(xdomain, scales)[0]
        shape = list(self.stage_shape(ldomain, ncomp, scales, 0))
        ax = self.dist.storage_order[pos]
        outer = int(np.prod(shape[:pos + 1]))
        inner = int(np.prod(shape[pos + 2:]))
        shape[pos + 1] = b.grid_size(scales[ax])
        out, out_d = ex.empty(tuple(shape)), ex.empty(tuple(shape))
        xb = self.stage_xb(ldomain, scales)
        if xb is not None and self.stage_xb(xdomain, scales) != xb:
            raise RuntimeError("dual z transform: the two domains disagree about the stage layout")
        kw = dict(xb=xb[0]) if (xb is not None and xb[0]) else {}
        if ctile:
            kw["tiled_row"] = ctile             # (src: a state field's rows as the solver keeps them)
        ex.transform_dual_z(spec, b, src, out, out_d, dvec, outer, inner, **kw)
        if self._needs_exchange(ldomain) and pos == 0:
            out = self._rows_after_z(ex, out, list(shape), blocked=xb is not None)
            out_d = self._rows_after_z(ex, out_d, list(shape), blocked=xb is not None)
        return out, out_d

    def backward_dual_step(self, domain, ncomp, src, scales, step, dscale):
        """Backward step `step` (a RealFourier axis, not the exchange step) applied to data that has seen the steps before
        it, twice from one read: -> (transform, transform of the derivative along that axis).  None when the executor
        has no dual kernel."""
        ex = self.dist.executor
        if not hasattr(ex, "transform_dual"):
            return None
        steps = self._steps(domain, scales)
        pos, b, spec = steps[step]
        if spec[0] != "rfft" or (self._needs_exchange(domain) and pos == 0):
            return None
        shape = list(self.stage_shape(domain, ncomp, scales, step))
        ax = self.dist.storage_order[pos]
        outer = int(np.prod(shape[:pos + 1]))
        inner = int(np.prod(shape[pos + 2:]))
        shape[pos + 1] = b.grid_size(scales[ax])
        out, out_d = ex.empty(tuple(shape)), ex.empty(tuple(shape))
        xb = self.stage_xb(domain, scales) if step == 1 else None
        kw = dict(xb=xb[1]) if xb is not None else {}
        win = self._window_of(src) if step == 1 else None
        if win is not None:
            _, gz, rows, rest = win["arrays"][src.data_ptr()]
            cw = gz // win["K"]
            z0 = win["k"] * cw
            shape[1] = cw
            out, out_d = ex.empty(tuple(shape)), ex.empty(tuple(shape))
            w = win["works"].pop((src.data_ptr(), win["k"]), None)
            if w is not None:
                w.wait()
            ex.transform_dual(spec, b, src, out, out_d, ncomp * cw, inner, dscale, xb=tuple(kw["xb"]) + (z0, cw))
            return out, out_d
        if self._x_step_by_component(src, ncomp):
            for c in range(ncomp):
                self.wait_for(src[c:c + 1])
                ex.transform_dual(spec, b, src[c:c + 1], out[c:c + 1], out_d[c:c + 1], outer // ncomp, inner, dscale, **kw)
        else:
            self.wait_for(src)
            ex.transform_dual(spec, b, src, out, out_d, outer, inner, dscale, **kw)
        return out, out_d

    def backward_data(self, domain, ncomp, c, g, scales, skip_last=False):
        """coefficient -> grid.  skip_last stops before the last storage axis ("pre-grid" layout)."""
        n = self.nsteps(domain, scales) - (1 if skip_last else 0)
        if n <= 0:
            self.dist.executor.copy(g, c)
            return
        self.backward_steps(domain, ncomp, c, scales, 0, n, dst=g)

    def forward_data(self, domain, ncomp, g, scales, c, skip_last=False, tiled_row=0):
        """tiled_row: the coefficient rows (the last transform's output, `c`) are written tile-major, rows of
        nx * tiled_row doubles (Executor.transform; only where `tiled_forward_ok` said so)."""
        self.forward_begin(domain, ncomp, g, scales, c, skip_last=skip_last, tiled_row=tiled_row)()

    def forward_windows(self, domain, ncomp, scales, c, tiled_row=0):
        """The forward transform of a product evaluated in windows of z planes (windows_begin): an object whose
        push(k, pg_k) runs the x step of window k ([ncomp][cw][Gx][..] pre-grid data) into the window's place of the
        exchanged layout and starts that window's parts of the pencil transpose, and whose finish() waits for all parts and
        runs the z step into `c`.  Pre-grid layout only (the last axis is the fused stage's): steps z, x."""
        ex = self.dist.executor
        steps = self._steps(domain, scales)[:-1]
        if [p for p, _, _ in steps] != [0, 1] or not self._needs_exchange(domain) or self.stage_xb(domain, scales) is None:
            raise RuntimeError("forward_windows: z / x steps around a blocked pencil transpose only")
        (_, bz, zspec), (_, bx, xspec) = steps
        xb = self.stage_xb(domain, scales)
        P = self.dist.size
        K = self._win["K"]
        full = list(self.pregrid_shape(domain, ncomp, scales, window=False))     # [ncomp][gz][Gx][rest..]
        gz, rest = full[1], int(np.prod(full[3:]))
        cw = gz // K
        nx = bx.coeff_size
        rows = nx // P
        send = ex.empty((ncomp, gz, nx) + tuple(full[3:]))          # blocked: [comp][p][gz][rows][rest]
        recv = ex.empty((ncomp, gz * P, rows) + tuple(full[3:]))    # [comp][Gz][nx_loc][rest]
        works = []
        tr = self

        class _Fwd:
            def push(self_, k, pg):
                z0 = k * cw
                ex.transform(xspec, bx, "forward", pg, send, ncomp * cw, rest, xb=tuple(xb[1]) + (z0, cw))
                works.append(tr.dist.pcomm.all_to_all_start(recv.reshape(-1), send.reshape(-1),
                                                            part=(z0 * rows * rest, cw * rows * rest), batch=ncomp))
                tr.dist.pcomm.stats["exchanges"] += ncomp

            def finish(self_):
                for w in works:
                    w.wait()
                del works[:]
                kw = dict(xb=xb[0]) if xb[0] else {}
                if tiled_row:
                    kw["tiled_row"] = tiled_row
                ex.transform(zspec, bz, "forward", recv, c, ncomp, rows * rest, **kw)
        return _Fwd()

    def forward_begin(self, domain, ncomp, g, scales, c, skip_last=False, tiled_row=0):
        """forward_data in two halves: runs the transforms up to the pencil transpose and STARTS it (blocked stage layout on
        several ranks: the components of the field as one group of sends / receives), and returns
        the function that waits for the arrivals and runs the remaining transforms.  A caller with several fields begins
        them all before it finishes the first (Solver.evaluate_F): the wire of one field runs under the x transforms of the
        next.  Without an exchange in flight everything happens in the first half and the returned function does nothing."""
        ex = self.dist.executor
        steps = self._steps(domain, scales)
        shape = [ncomp] + list(domain.storage_grid_shape(scales))
        if skip_last:
            steps = steps[:-1]
            shape = list(self.pregrid_shape(domain, ncomp, scales))
        if not steps:
            ex.copy(c, g)
            return lambda: None
        exchange = self._needs_exchange(domain)
        rsteps = list(reversed(steps))
        blocked = exchange and self.stage_xb(domain, scales) is not None
        defer = blocked and os.environ.get("DDH_A2A_DEFER", "1") != "0"
        split_x = os.environ.get("DDH_A2A_SPLIT_X", "0") == "1"
        state = dict(src=g, works=None, second_half=False)

        def run(i_from, i_to):
            src = state["src"]
            for i in range(i_from, i_to):
                pos, b, spec = rsteps[i]
                if exchange and pos == 0:
                    # [comp, Gz/P, nx, ny] -> [comp, Gz, nx_loc, ny]
                    P = self.dist.size
                    nc, Gzl, nx = shape[0], shape[1], shape[2]
                    rest = int(np.prod(shape[3:]))
                    if state["works"] is None:          # (else: started by the component-wise x step below)
                        tmp = ex.empty((nc, Gzl * P, nx // P) + tuple(shape[3:]))
                        works = []
                        if blocked:
                            # blocked x side: the x transform has written [comp][p][Gz / P][nx_loc][..] -- a component is sent
                            # as it lies and received as the component [Gz][nx_loc][..]: no kernel at all
                            works = [self.dist.pcomm.all_to_all_start(tmp.reshape(-1), src.reshape(-1),
                                                                      part=(0, Gzl * (nx // P) * rest), batch=nc)]
                            self.dist.pcomm.stats["exchanges"] += nc
                        elif nc > 1 and _overlap():
                            n1 = Gzl * nx * rest
                            pend = None
                            for cc in range(nc):
                                send = ex.empty((n1,))
                                ex.a2a_pack(src[cc:cc + 1], send, Gzl, nx, 1, rest, P)
                                # the received order [p][Gz / P][nx_loc][rest] IS the component [Gz][nx_loc][rest]: received in
                                # place (round 5 unpacked with a kernel that copied the buffer onto the component unchanged)
                                recv = tmp[cc:cc + 1].reshape(-1)
                                work = self.dist.pcomm.all_to_all_start(recv, send)
                                if pend is not None:
                                    pend[0].wait()
                                pend = (work, recv, cc, send)
                            pend[0].wait()
                        else:
                            self._exchange(ex, "columns", src, tmp, nc, Gzl * P, nx, rest)
                        state["works"] = (tmp, works)
                    if defer and not state["second_half"]:
                        state["src"] = src              # the exchanges are in flight: the second half re-enters here
                        return i
                    tmp, works = state["works"]
                    state["works"] = None
                    for w in works:
                        w.wait()
                    shape[1], shape[2] = Gzl * P, nx // P
                    src = tmp
                outer = int(np.prod(shape[:pos + 1]))
                inner = int(np.prod(shape[pos + 2:]))
                shape[pos + 1] = b.coeff_size
                last = i == len(rsteps) - 1
                dst = c if last else ex.empty(tuple(shape))
                xb = self.stage_xb(domain, scales) if pos <= 1 else None
                kw = dict(xb=xb[pos]) if (xb is not None and xb[pos]) else {}  # step 1 (x) writes, step 0 (z) reads the stage-1 array
                if last and tiled_row:
                    kw["tiled_row"] = tiled_row
                if defer and split_x and pos == 1 and i + 1 < len(rsteps) and rsteps[i + 1][0] == 0:
                    # the x step before a deferred blocked exchange: component by component, each one's exchange started
                    # behind its transform -- component c is on the wire while component c + 1 is transformed
                    P = self.dist.size
                    nc = shape[0]
                    tmp = ex.empty((nc, shape[1] * P, shape[2] // P) + tuple(shape[3:]))
                    works = []
                    for cc in range(nc):
                        ex.transform(spec, b, "forward", src[cc:cc + 1], dst[cc:cc + 1], outer // nc, inner, **kw)
                        works.append(self.dist.pcomm.all_to_all_start(tmp[cc:cc + 1].reshape(-1), dst[cc:cc + 1].reshape(-1)))
                    self.dist.pcomm.stats["exchanges"] += nc
                    state["works"] = (tmp, works)
                else:
                    ex.transform(spec, b, "forward", src, dst, outer, inner, **kw)
                src = dst
            state["src"] = src
            return i_to

        split = run(0, len(rsteps))             # returns early, at the exchange step, when that one was only started

        def finish():
            state["second_half"] = True
            run(split, len(rsteps))
        return finish
