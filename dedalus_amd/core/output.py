"""
Analysis output: scheduled task handlers, HDF5 analysis sets and restarts.

Mirrors the reference's `solver.evaluator` (core/evaluator.py:30-700) for the three field systems of this package
(Cartesian, sphere, shell):

* `Handler` scheduling (`sim_dt`, `wall_dt`, `iter`, `custom_schedule`, groups) with the reference's rule for
  simulation-time cadences (evaluator.py:236-258); handlers are checked at the START of a step, on the pre-step state,
  as the reference's timesteppers do.
* `FileHandler`: the reference's on-disk format (`<base>/<base>_s<N>.h5`; `scales/{sim_time, timestep, wall_time,
  iteration, write_number, constant, <coord>_hash_<sha1>}`, `tasks/<name>` of shape (writes, *tensor, *space) with the
  `constant / grid_space / scales` attributes, dimension labels and attached dimension scales, evaluator.py:509-607),
  written by tools/h5lite.py (no h5py / libhdf5 in this image).  Device data reach the file through the fields' host
  mirrors (`field['g']` / `field['c']`: one device-to-host copy per task and write); on several ranks the local blocks
  are gathered to rank 0, which writes (the reference's 'gather' mode, evaluator.py:609-645).
* `DictionaryHandler`, `add_tasks`, and `load_state` for the IVP solvers (core/solvers.py:632-673).
"""

import hashlib
import os
import pathlib
import re
import shutil
import time
import weakref

import numpy as np

from ..tools import h5lite


class Handler:
    """Group of tasks with an evaluation schedule (core/evaluator.py:206-330)."""

    def __init__(self, solver, group=None, wall_dt=None, sim_dt=None, iter=None, custom_schedule=None):
        self.solver, self.dist = solver, solver.dist
        self.group = group
        self.wall_dt, self.sim_dt, self.iter, self.custom_schedule = wall_dt, sim_dt, iter, custom_schedule
        self.tasks = []
        # initial divisors of -1 trigger output on the first iteration
        self.last_wall_div = self.last_sim_div = self.last_iter_div = -1

    def check_schedule(self, **kw):
        scheduled = False
        if self.wall_dt:
            wall_div = kw['wall_time'] // self.wall_dt
            if wall_div > self.last_wall_div:
                scheduled, self.last_wall_div = True, wall_div
        if self.sim_dt:
            # output if the target closest to the current time has not triggered yet and the next step does not get closer
            t, dt = kw['sim_time'], kw['timestep']
            closest = int(np.round(t / self.sim_dt))
            if closest > self.last_sim_div:
                target = closest * self.sim_dt
                if abs(t - target) < abs(t + dt - target):
                    scheduled, self.last_sim_div = True, closest
        if self.iter:
            iter_div = kw['iteration'] // self.iter
            if iter_div > self.last_iter_div:
                scheduled, self.last_iter_div = True, iter_div
        if self.custom_schedule:
            if self.custom_schedule(**kw):
                scheduled = True
        return scheduled

    def add_task(self, task, layout='g', name=None, scales=None):
        if name is None:
            name = str(task)
        expr = self.solver.problem._parse(task) if isinstance(task, str) else task
        layout = 'c' if layout in ('c', 'coeff') else 'g'
        self.tasks.append(dict(operator=expr, layout=layout, name=name, scales=scales, out=None))

    def add_tasks(self, tasks, **kw):
        name = kw.pop('name', '')
        for task in tasks:
            self.add_task(task, name=name + str(task), **kw)

    def add_system(self, system, **kw):
        self.add_tasks(getattr(system, "fields", system), **kw)

    def evaluate(self):
        for task in self.tasks:
            op = task['operator']
            out = op.evaluate() if hasattr(op, "evaluate") else op
            # every output passes through coefficient space before it is written (dealiasing, evaluator.py:157-160)
            if hasattr(out, "require_coeff_space"):
                out.require_coeff_space()
            if hasattr(out, "change_scales"):
                out.change_scales(1 if task['scales'] is None else task['scales'])     # remedy_scales(None) = 1
            task['out'] = out

    def process(self, **kw):
        raise NotImplementedError


class DictionaryHandler(Handler):
    """Handler that keeps the evaluated fields in a dictionary (core/evaluator.py:333-347)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.fields = {}

    def __getitem__(self, item):
        return self.fields[item]

    def process(self, **kw):
        for task in self.tasks:
            self.fields[task['name']] = task['out']


# ---- per-system description of a field's user array ------------------------------------------------------------------

def _scale_tuple(scales, n):
    if scales is None:
        return (1.0,) * n
    if np.isscalar(scales):
        return (float(scales),) * n
    return tuple(float(s) for s in scales)


def _describe(dist, field, layout, data):
    """-> (axes, constant flags, scales, shard axis) with axes = [(scale name, coordinate data or None)] per spatial axis
    of the GLOBAL user array and shard axis = index (among ALL axes of `data`) along which ranks hold blocks"""
    from . import shell as _shell, sphere as _sphere
    grid = layout == 'g'
    if isinstance(field, (_shell.ShellField, _sphere.SField, _shell.ShAzimuthalInterp)):
        basis = field.basis
        names = [c.name if hasattr(c, "name") else str(c) for c in dist.coords]
        dim = len(names)
        scales = _scale_tuple(field.scales, dim)
        rank = field.rank
        if grid:
            grids = basis.grids(scales)
            axes = [(names[i], np.asarray(grids[i], dtype=np.float64).ravel()) for i in range(dim)]
        else:
            axes = []
            for i in range(dim):
                n = data.shape[rank + i] * (dist.size if (i == 0 and dist.size > 1) else 1)
                axes.append(('k' + names[i], (np.arange(n) // 2 if i == 0 else np.arange(n)).astype(np.int64)))
        const = [False] * dim
        if isinstance(basis, _shell.SurfaceBasis):
            const[2] = True
            axes[2] = ('constant', None)
        for i in getattr(field, "const_axes", ()):
            const[i] = True
            axes[i] = ('constant', None)
        shard = None
        if dist.size > 1:
            shard = rank + (1 if grid else 0)
        return axes, const, scales, shard
    # Cartesian
    dom = field.domain
    dim = dist.dim
    scales = _scale_tuple(getattr(field, "scales", None), dim)
    rank = len(field.tensorsig)
    axes, const = [], []
    for ax in range(dim):
        b = dom.by_axis[ax]
        cname = dist.coords[ax].name
        if b is None:
            axes.append(('constant', None))
            const.append(True)
            continue
        const.append(False)
        if grid:
            axes.append((cname, np.asarray(b.global_grid(scales[ax]), dtype=np.float64).ravel()))
        else:
            n = b.coeff_size if hasattr(b, "coeff_size") else b.size
            groups = np.arange(n) // 2 if type(b).__name__ in ("RealFourier",) else np.arange(n)
            axes.append(('k' + cname, groups.astype(np.int64)))
    shard = None
    if dist.size > 1:
        ax = dist.shard_grid_axis if grid else dist.shard_coeff_axis
        if ax is not None and dom.by_axis[ax] is not None:
            shard = rank + ax
    return axes, const, scales, shard


class FileHandler(Handler):
    """Handler that writes its tasks to HDF5 analysis sets (H5FileHandlerBase / H5GatherFileHandler of the reference)."""

    def __init__(self, base_path, solver, max_writes=None, mode=None, parallel=None, **kw):
        super().__init__(solver, **kw)
        if parallel not in (None, 'gather'):
            raise NotImplementedError("file handler parallel mode %r (rank 0 gathers and writes)" % (parallel,))
        mode = (mode or "overwrite").lower()
        if mode not in ("overwrite", "append"):
            raise ValueError("Write mode {} not defined.".format(mode))
        base_path = pathlib.Path(base_path).resolve()
        if base_path.is_file():
            raise ValueError("base_path should indicate a folder for storing HDF5 files.")
        self.base_path, self.name, self.max_writes = base_path, base_path.stem, max_writes
        self.is_root = getattr(self.dist, "rank", 0) == 0
        self.set_num, self.total_write_num = 1, 0
        if self.is_root:
            sets = list(base_path.glob("%s_s*" % self.name))
            if mode == "overwrite":
                for s in sets:
                    shutil.rmtree(str(s)) if s.is_dir() else s.unlink()
            elif sets:
                nums = [int(m.group(1)) for m in (re.match(r"%s_s(\d+)$" % re.escape(self.name), s.stem) for s in sets) if m]
                last = base_path.joinpath("%s_s%d.h5" % (self.name, max(nums)))
                wn = h5lite.read(str(last))['scales/write_number'].read()
                self.set_num, self.total_write_num = max(nums) + 1, (int(wn[-1]) if wn.size else 0)
            base_path.mkdir(exist_ok=True, parents=True)
        pc = getattr(self.dist, "pcomm", None)
        if pc is not None:
            self.set_num = int(pc.allreduce_max(self.set_num if self.is_root else 0))
            self.total_write_num = int(pc.allreduce_max(self.total_write_num if self.is_root else 0))
        self.file_write_num = 0
        self._file = None
        self._dsets = None
        self._queue = []
        # asynchronous device-to-host staging of the task data (DDH_OUTPUT_SYNC=1: fetch and write immediately)
        self.async_staging = os.environ.get("DDH_OUTPUT_SYNC", "0") != "1"
        # a script that never closes its handlers still gets its last write: flushed when the main loop ends
        # (IVPLifecycle) and, as the last resort, at interpreter exit: atexit holds the handler itself (a finalizer of the
        # handler could not work -- by the time it runs the handler and its staged outputs are gone -- and a handler that
        # a script drops must still write what it staged), and the exit path CLOSES the set (flush + close of the file)
        self._exit_hook = False                      # registered while outputs may be pending (process), dropped by close()

    def _at_exit(self):
        try:
            self.close()
        except Exception:                           # the device may already be gone at interpreter exit
            pass

    @property
    def current_file(self):
        return self.base_path.joinpath("%s_s%d.h5" % (self.name, self.set_num))

    # ---- file creation: everything is declared from the first evaluated outputs -------------------------------------------
    def _create_file(self, gathered):
        f = h5lite.File(str(self.current_file))
        f.attrs['set_number'] = int(self.set_num)
        f.attrs['handler_name'] = self.name
        f.attrs['writes'] = 0
        sc = f.create_group('scales')
        const = sc.create_dataset('constant', data=np.zeros(1))
        const.make_scale('constant')
        tsc = {}
        for n in ('sim_time', 'timestep', 'wall_time'):
            tsc[n] = sc.create_dataset(n, shape=(0,), maxshape=(None,), dtype=np.float64)
            tsc[n].make_scale(n)
        for n in ('iteration', 'write_number'):
            tsc[n] = sc.create_dataset(n, shape=(0,), maxshape=(None,), dtype=np.int64)
            tsc[n].make_scale(n)
        tk = f.create_group('tasks')
        dsets = {}
        for task, (data, axes, const_flags, scales) in zip(self.tasks, gathered):
            d = tk.create_dataset(task['name'], shape=(0,) + data.shape, maxshape=(None,) + data.shape, dtype=data.dtype)
            d.attrs['constant'] = np.array(const_flags, dtype=bool)
            d.attrs['grid_space'] = np.array([task['layout'] == 'g'] * len(axes), dtype=bool)
            d.attrs['scales'] = np.array(scales, dtype=np.float64)
            d.set_label(0, 't')
            for n in ('sim_time', 'wall_time', 'timestep', 'iteration', 'write_number'):
                d.attach_scale(0, tsc[n])
            rank = data.ndim - len(axes)
            for i, (sn, coord) in enumerate(axes):
                if coord is None:
                    scale = const
                else:
                    lookup = "%s_hash_%s" % (sn, hashlib.sha1(np.ascontiguousarray(coord)).hexdigest())
                    if lookup not in sc:
                        sc.create_dataset(lookup, data=coord, dtype=coord.dtype).make_scale(sn)
                    scale = sc[lookup]
                d.set_label(1 + rank + i, sn)
                d.attach_scale(1 + rank + i, scale)
            dsets[task['name']] = d
        f.commit()
        self._file, self._dsets, self._tsc = f, dsets, tsc

    def _gather(self, task):
        out = task['out']
        data = np.asarray(out[task['layout']])
        axes, const, scales, shard = _describe(self.dist, out, task['layout'], data)
        pc = getattr(self.dist, "pcomm", None)
        if pc is not None and shard is not None:
            data = pc.all_gather_host(np.ascontiguousarray(data), axis=shard)
        return np.array(data), axes, const, scales

    def _start(self, task):
        """Begin fetching a task's data: device fields on one rank are staged asynchronously (device snapshot + pinned
        host copy on a side stream, Field.snapshot_async) and resolved when the write is flushed -- at the next
        scheduled output or at close -- so the copy overlaps the timesteps in between.  Returns a callable ->
        (data, axes, constant flags, scales)."""
        out = task['out']
        pc = getattr(self.dist, "pcomm", None)
        if self.async_staging and pc is None and hasattr(out, "snapshot_async") and hasattr(out, "domain"):
            pend = out.snapshot_async(task['layout'])
            if pend is not None:
                axes, const, scales, _ = _describe(self.dist, out, task['layout'], None)
                self._staged = True
                return lambda: (pend(), axes, const, scales)
        res = self._gather(task)
        return lambda: res

    def process(self, iteration=0, wall_time=0.0, sim_time=0.0, timestep=0.0, **kw):
        if not self._exit_hook:
            import atexit
            atexit.register(self._at_exit)
            self._exit_hook = True
        self.flush()                                   # the previous write: its copies had a whole output interval
        self.total_write_num += 1
        meta = dict(sim_time=sim_time, wall_time=wall_time, timestep=timestep, iteration=iteration,
                    write_number=self.total_write_num)
        self._staged = False
        self._queue.append((meta, [self._start(task) for task in self.tasks]))
        if not self._staged:
            self.flush()                               # nothing in flight (host data, gathered ranks): write now

    def flush(self):
        """Write every staged output to the analysis set (in order)."""
        while self._queue:
            meta, entries = self._queue.pop(0)
            self._write(meta, [e() for e in entries])

    def _write(self, meta, gathered):
        self.file_write_num += 1
        roll = self.max_writes is not None and self.file_write_num > self.max_writes
        if self._dsets is not None and any(d.nrows >= d.capacity for d in self._dsets.values()):
            roll = True
        if roll:
            self._close_file()
            self.set_num += 1
            self.file_write_num = 1
        if not self.is_root:
            return
        if self._file is None:
            self._create_file(gathered)
        f = self._file
        for n, v in meta.items():
            self._tsc[n].append(v)
        for task, (data, _, _, _) in zip(self.tasks, gathered):
            self._dsets[task['name']].append(data)
        f.set_scalar_attr(f, 'writes', self.file_write_num)
        f.flush()

    def close(self):
        self.flush()
        self._close_file()
        if self._exit_hook:                          # nothing pending any more: do not pin the handler (solver, device
            import atexit                            # buffers) until interpreter exit
            atexit.unregister(self._at_exit)
            self._exit_hook = False

    def _close_file(self):
        if self._file is not None:
            self._file.close()
        self._file = self._dsets = None


class OutputEvaluator:
    """`solver.evaluator`: handler registry and scheduler (core/evaluator.py:30-120)."""

    def __init__(self, solver):
        self.solver = solver
        self.handlers = []
        self.groups = {}

    def add_handler(self, handler):
        self.handlers.append(handler)
        if handler.group is not None:
            self.groups.setdefault(handler.group, []).append(handler)
        return handler

    def add_dictionary_handler(self, **kw):
        return self.add_handler(DictionaryHandler(self.solver, **kw))

    def add_file_handler(self, filename, parallel=None, **kw):
        return self.add_handler(FileHandler(filename, self.solver, parallel=parallel, **kw))

    def evaluate_group(self, group, **kw):
        self.evaluate_handlers(self.groups.get(group, []), **kw)

    def evaluate_scheduled(self, **kw):
        self.evaluate_handlers([h for h in self.handlers if h.check_schedule(**kw)], **kw)

    def evaluate_handlers(self, handlers=None, **kw):
        handlers = self.handlers if handlers is None else handlers
        for h in handlers:
            h.evaluate()
            h.process(**kw)

    def flush(self):
        """Write every output that is still staged (asynchronous device-to-host copies) to its file.  Called when a
        run ends -- `proceed` turning False, `evolve` returning or raising, `log_stats` -- and at interpreter exit, so a
        reference-style script (`while solver.proceed: solver.step(dt)`) that never closes its handlers loses nothing:
        the reference writes inside `process()` (core/evaluator.py:366-700)."""
        for h in self.handlers:
            fl = getattr(h, "flush", None)
            if fl is not None:
                fl()

    # hook called by the IVP solvers at the start of every step (pre-step state, as the reference's timesteppers do)
    def step_hook(self, solver):
        if not self.handlers:
            return
        wall = getattr(solver, "_step_wall_time", None)           # the world clock of this step (same on every rank)
        if wall is None:
            wall = time.time() - solver.start_time
        # the timestep of THIS step (core/timesteppers.py:150, 608), not the previous one's
        dt = getattr(solver, "_step_dt", None)
        self.evaluate_scheduled(iteration=int(solver.iteration), wall_time=wall,
                                sim_time=float(solver.sim_time), timestep=float(solver.dt if dt is None else dt))


def load_state(solver, path, index=-1, allow_missing=False):
    """Load the state of an IVP solver from an analysis set (core/solvers.py:632-673, Field.load_from_hdf5
    core/field.py:815-843) -> (write number, timestep).  On several ranks every rank reads the file and keeps its block."""
    import logging
    logger = logging.getLogger("solvers")
    r = h5lite.read(str(path))
    write = int(r['scales/write_number'].read()[index])
    dt = float(r['scales/timestep'].read()[index])
    solver.iteration = solver.initial_iteration = int(r['scales/iteration'].read()[index])
    solver.sim_time = solver.initial_sim_time = float(r['scales/sim_time'].read()[index])
    logger.info("Loading solver state from: {}".format(path))
    logger.info("Loading iteration: {}".format(solver.iteration))
    logger.info("Loading write: {}".format(write))
    logger.info("Loading sim time: {}".format(solver.sim_time))
    logger.info("Loading timestep: {}".format(dt))
    tasks = r['tasks']
    fields = getattr(solver, "state", None) or solver.variables
    for field in fields:
        name = getattr(field, "name", None)
        if name in tasks:
            d = tasks[name]
            grid = bool(np.all(d.attrs.get('grid_space', True)))
            layout = 'g' if grid else 'c'
            scales = tuple(float(s) for s in np.atleast_1d(d.attrs.get('scales', 1.0)))
            data = d.read(index)
            if hasattr(field, "change_scales"):
                field.change_scales(scales if len(scales) > 1 else scales[0])
            if getattr(solver.dist, "size", 1) > 1:
                _, _, _, shard = _describe(solver.dist, field, layout, np.asarray(field[layout]))
                if shard is not None:
                    n = data.shape[shard] // solver.dist.size
                    sl = [slice(None)] * data.ndim
                    sl[shard] = slice(solver.dist.rank * n, (solver.dist.rank + 1) * n)
                    data = data[tuple(sl)]
            field[layout] = data
        elif allow_missing:
            logger.warning("Field '%s' not found in savefile." % name)
        else:
            raise IOError("Field '%s' not found in savefile. Set allow_missing=True to ignore this error." % name)
    return write, dt
