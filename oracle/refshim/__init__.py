"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Reference shim: import and run the *unmodified* reference (Dedalus v3,
``/root/reference``) inside this container, where mpi4py / FFTW / h5py / numexpr
are not installed and no network exists.

Nothing is copied from the reference.  The reference package is imported from where
it lies (``sys.path`` gets ``/root/reference``) and only the modules it cannot import
here are pre-seeded in ``sys.modules`` with single-process stand-ins:

* ``mpi4py.MPI``                              size-1 communicator (API surface: SURVEY.md section 2.3)
* ``dedalus.libraries.fftw.fftw_wrappers``    scipy.fft (pocketfft) behind the names of
                                              ``dedalus/libraries/fftw/fftw_wrappers.pyx:28-352``
* ``dedalus.core.transposes``                 names only (never constructed with one rank,
                                              ``dedalus/core/distributor.py:98,134,146-165``)
* ``dedalus.tools.linalg``, ``dedalus.libraries.spin_recombination``
                                              the reference's own Cython, compiled by
                                              ``oracle/build_ref.py`` into ``oracle/_ref/`` when
                                              available, else scipy stand-ins
* ``h5py``, ``numexpr``                       import stubs

This only works where ``/root/reference`` exists (the build container).  The GPU box has
no reference: tests there use the committed fixtures in ``tests/golden/`` produced by
``oracle/make_golden.py`` through this shim.
"""

import importlib.machinery
import importlib.util
import os
import sys
import time
import types

import numpy as np

REFERENCE_PATH = os.environ.get("DEDALUS_REFERENCE_PATH", "/root/reference")
_REF_BUILD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_ref")


def available():
    return os.path.isdir(os.path.join(REFERENCE_PATH, "dedalus"))


# --------------------------------------------------------------------------------------
# mpi4py stand-in
# --------------------------------------------------------------------------------------

def _make_mpi():
    mpi4py = types.ModuleType("mpi4py")
    MPI = types.ModuleType("mpi4py.MPI")
    MPI.SUM, MPI.MAX, MPI.MIN, MPI.DOUBLE, MPI.IN_PLACE = "SUM", "MAX", "MIN", "DOUBLE", "IN_PLACE"

    class Comm:
        size = 1
        rank = 0
        dim = 0
        coords = ()

        def Get_size(self):
            return 1

        def Get_rank(self):
            return 0

        def Create_cart(self, dims, **kw):
            c = Comm()
            c.dim = len(dims)
            c.coords = tuple(0 for _ in dims)
            return c

        def Sub(self, remain):
            return Comm()

        def Get_coords(self, rank):
            return list(self.coords)

        def Barrier(self):
            pass

        def bcast(self, x, root=0):
            return x

        def Bcast(self, x, root=0):
            pass

        def allreduce(self, x, op=None):
            return x

        def Allreduce(self, s, r, op=None):
            if not (isinstance(s, str) and s == "IN_PLACE"):
                r[...] = s

        def reduce(self, x, op=None, root=0):
            return x

        def gather(self, x, root=0):
            return [x]

        def allgather(self, x):
            return [x]

        def scatter(self, x, root=0):
            return x[0]

    MPI.Comm = Comm
    MPI.COMM_WORLD = Comm()
    MPI.COMM_SELF = Comm()
    MPI.Wtime = time.time
    mpi4py.MPI = MPI
    return mpi4py, MPI


# --------------------------------------------------------------------------------------
# FFTW wrapper stand-in (pocketfft).  Conventions: fftw_wrappers.pyx:61-214 (unnormalised
# both ways, r2c output length N//2+1) and :217-334 (REDFT10 / REDFT01 unnormalised).
# --------------------------------------------------------------------------------------

def _make_fftw():
    import scipy.fft as sf
    m = types.ModuleType("dedalus.libraries.fftw.fftw_wrappers")
    m.fftw_flags = {}
    m.fftw_mpi_init = lambda: None
    m.create_buffer = lambda n: np.zeros(int(n))
    m.create_array = lambda shape, dtype: np.zeros(tuple(int(s) for s in shape), dtype=dtype)
    m.create_copy = lambda a: np.array(a, copy=True)

    class FourierTransform:
        def __init__(self, dtype, gshape, axis, flags=()):
            self.real = np.dtype(dtype) == np.float64
            self.axis = axis
            self.N = gshape[axis]
            cs = list(gshape)
            if self.real:
                cs[axis] = gshape[axis] // 2 + 1
            self.cshape = np.array(cs, dtype=int)

        def forward(self, g, c):
            c[...] = sf.rfft(g, axis=self.axis) if self.real else sf.fft(g, axis=self.axis)

        def backward(self, c, g):
            if self.real:
                g[...] = sf.irfft(c, n=self.N, axis=self.axis) * self.N
            else:
                g[...] = sf.ifft(c, axis=self.axis) * self.N

    class DiscreteCosineTransform:
        def __init__(self, dtype, gshape, axis, flags=()):
            self.axis = axis

        def forward(self, g, c):
            c[...] = sf.dct(g, type=2, axis=self.axis)

        def backward(self, c, g):
            g[...] = sf.dct(c, type=3, axis=self.axis)

    class DiscreteSineTransform:
        def __init__(self, dtype, gshape, axis, flags=()):
            self.axis = axis

        def forward(self, g, c):
            c[...] = sf.dst(g, type=2, axis=self.axis)

        def backward(self, c, g):
            g[...] = sf.dst(c, type=3, axis=self.axis)

    class R2HCTransform:
        def __init__(self, *a, **k):
            raise RuntimeError("refshim: fftw_hc is not available (no FFTW)")

    m.FourierTransform = FourierTransform
    m.DiscreteCosineTransform = DiscreteCosineTransform
    m.DiscreteSineTransform = DiscreteSineTransform
    m.R2HCTransform = R2HCTransform
    return m


def _make_transposes():
    m = types.ModuleType("dedalus.core.transposes")

    class _NoTranspose:
        def __init__(self, *a, **k):
            raise RuntimeError("refshim: single process, no distributed transposes")

    m.FFTWTranspose = m.AlltoallvTranspose = m.RowDistributor = m.ColDistributor = _NoTranspose
    return m


# --------------------------------------------------------------------------------------
# Cython units: the reference's own compiled code if oracle/build_ref.py produced it,
# otherwise scipy equivalents of dedalus/tools/linalg.pyx:20-357.
# --------------------------------------------------------------------------------------

def _load_ext(fullname, stem):
    if not os.path.isdir(_REF_BUILD):
        return None
    for fn in os.listdir(_REF_BUILD):
        if fn.startswith(stem + ".") and fn.endswith(".so"):
            path = os.path.join(_REF_BUILD, fn)
            loader = importlib.machinery.ExtensionFileLoader(fullname, path)
            spec = importlib.util.spec_from_loader(fullname, loader, origin=path)
            mod = importlib.util.module_from_spec(spec)
            loader.exec_module(mod)
            return mod
    return None


def _make_linalg_fallback():
    from scipy import sparse
    from scipy.sparse.linalg import spsolve_triangular
    m = types.ModuleType("dedalus.tools.linalg")

    def _csr(shape_rows, indptr, indices, data, ncols):
        return sparse.csr_matrix((data, indices, indptr), shape=(shape_rows, ncols))

    def _apply(indptr, indices, data, arr, out, axis):
        nrows = len(indptr) - 1
        ncols = arr.shape[axis]
        A = _csr(nrows, indptr, indices, data, ncols)
        a = np.moveaxis(arr, axis, 0)
        o = np.moveaxis(out, axis, 0)
        o[...] = (A @ a.reshape(ncols, -1)).reshape(o.shape)

    def _solve(indptr, indices, data, arr, axis):
        n = len(indptr) - 1
        A = _csr(n, indptr, indices, data, n)
        a = np.moveaxis(arr, axis, 0)
        a[...] = spsolve_triangular(A.tocsr(), a.reshape(n, -1).copy(), lower=False).reshape(a.shape)

    m.apply_csr_vec = lambda ip, ix, d, a, o, nt: _apply(ip, ix, d, a, o, 0)
    m.apply_csr_first = lambda ip, ix, d, a, o, nt: _apply(ip, ix, d, a, o, 0)
    m.apply_csr_last = lambda ip, ix, d, a, o, nt: _apply(ip, ix, d, a, o, 1)
    m.apply_csr_mid = lambda ip, ix, d, a, o, nt: _apply(ip, ix, d, a, o, 1)
    m.solve_upper_csr_vec = lambda ip, ix, d, a, nt: _solve(ip, ix, d, a, 0)
    m.solve_upper_csr_first = lambda ip, ix, d, a, nt: _solve(ip, ix, d, a, 0)
    m.solve_upper_csr_last = lambda ip, ix, d, a, nt: _solve(ip, ix, d, a, 1)
    m.solve_upper_csr_mid = lambda ip, ix, d, a, nt: _solve(ip, ix, d, a, 1)
    return m


def _make_spin_fallback():
    m = types.ModuleType("dedalus.libraries.spin_recombination")

    def _nyi(*a, **k):
        raise RuntimeError("refshim: spin_recombination needs oracle/build_ref.py")

    m.recombine_forward = m.recombine_backward = _nyi
    return m


_loaded = None


def load_reference(quiet=True):
    """Return the reference's ``dedalus.public`` module (imported unmodified)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference not found at %s (refshim only works in the build container)"
                           % REFERENCE_PATH)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("NUMEXPR_MAX_THREADS", "1")
    mpi4py, MPI = _make_mpi()
    sys.modules.setdefault("mpi4py", mpi4py)
    sys.modules.setdefault("mpi4py.MPI", MPI)
    for name in ("h5py", "numexpr"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except ImportError:
                stub = types.ModuleType(name)
                if name == "h5py":
                    class File:
                        def __init__(self, *a, **k):
                            raise RuntimeError("refshim: h5py stub")
                    stub.File = File
                else:
                    def evaluate(*a, **k):
                        raise RuntimeError("refshim: numexpr stub")
                    stub.evaluate = evaluate
                sys.modules[name] = stub
    sys.modules["dedalus.libraries.fftw.fftw_wrappers"] = _make_fftw()
    sys.modules["dedalus.core.transposes"] = _make_transposes()
    sys.modules["dedalus.tools.linalg"] = (_load_ext("dedalus.tools.linalg", "linalg")
                                           or _make_linalg_fallback())
    sys.modules["dedalus.libraries.spin_recombination"] = (
        _load_ext("dedalus.libraries.spin_recombination", "spin_recombination")
        or _make_spin_fallback())
    if REFERENCE_PATH not in sys.path:
        sys.path.insert(0, REFERENCE_PATH)
    import logging
    import dedalus.public as d3
    if quiet:
        logging.getLogger().setLevel(logging.WARNING)
        for name in list(logging.root.manager.loggerDict):
            if name.startswith("dedalus") or name in ("subsystems", "solvers", "problems", "distributor"):
                logging.getLogger(name).setLevel(logging.WARNING)
    # make the stand-ins reachable as attributes of their parent packages
    import dedalus.libraries.fftw as _f
    _f.fftw_wrappers = sys.modules["dedalus.libraries.fftw.fftw_wrappers"]
    _loaded = d3
    return d3
