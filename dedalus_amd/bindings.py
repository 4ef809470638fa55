"""
Reference-side bindings of libdedalus_hip.so: the plan classes a Dedalus maintainer registers at the reference's own
plugin points.  They need nothing but the C ABI (ctypes table in libhip.py) and
keep the reference's constructor signatures and method names:

    B1  register_transform(basis_cls, name)          core/transforms.py:27-32
          HipRealFFT(grid_size, coeff_size)                                          :371, 537-565
          HipComplexFFT(grid_size, coeff_size)                                       :194, 292-330
          HipJacobi(grid_size, coeff_size, a, b, a0, b0, dealias_before_converting)  :102, 801-902, 114-158
          HipSWSHColatitude(Ntheta, Lmax, m_maps, s)                                 :1254-1340
        methods forward(gdata, cdata, axis) / backward(cdata, gdata, axis) (:43-51) on C-contiguous views that MAY
        ALIAS ONE BUFFER (the reference's Field hands out grid and coefficient views of the same memory,
        core/basis.py:185-193): the library entry points are in-place safe.
    B3  TransposePlanner(global_shape, chunk_shape, dtype, axis, comm)               core/distributor.py:26-29, 740
          HipTranspose: localize_rows(CL, RL) / localize_columns(RL, CL)             core/transposes.pyx:248-266

Arrays are anything that exposes a device pointer -- torch tensors (`data_ptr()`), cupy arrays (`.data.ptr`), objects
with `__cuda_array_interface__` -- or host NumPy arrays, which are staged through ddh_memcpy_h2d / d2h (PCIe cost;
useful to try the library from an unmodified host-resident reference).

`install(register_transform, RealFourier=..., ComplexFourier=..., Jacobi=..., SphereBasis=..., name='hip')` performs
the registrations; tests/test_gpu_boundary.py drives it with a stand-in registry of the same shape (the GPU box has no
reference checkout).
"""

import ctypes as C
import math

import numpy as np

from . import libhip


# ---- array access ------------------------------------------------------------------------------------------------
def device_pointer(a):
    """Device address of an array-like, or None for host NumPy data."""
    if isinstance(a, np.ndarray):
        return None
    if hasattr(a, "data_ptr"):
        return int(a.data_ptr())
    if hasattr(a, "__cuda_array_interface__"):
        return int(a.__cuda_array_interface__["data"][0])
    if hasattr(a, "data") and hasattr(a.data, "ptr"):
        return int(a.data.ptr)
    raise TypeError("cannot find a device pointer in %r" % type(a))


class _Staged:
    """Run fn(in_ptr, out_ptr) on device copies of host arrays (same-buffer host views keep their aliasing)."""

    def __init__(self, src, dst):
        self.src, self.dst = src, dst

    def run(self, fn):
        ps, pd = device_pointer(self.src), device_pointer(self.dst)
        if ps is not None and pd is not None:
            return fn(C.c_void_p(ps), C.c_void_p(pd))
        if ps is not None or pd is not None:
            raise TypeError("grid and coefficient data must both be device arrays or both host arrays")
        src = np.ascontiguousarray(self.src)
        out = np.empty(self.dst.shape, dtype=self.dst.dtype)
        d_in, d_out = C.c_void_p(), C.c_void_p()
        libhip.call("ddh_alloc", C.byref(d_in), max(src.nbytes, 16))
        libhip.call("ddh_alloc", C.byref(d_out), max(out.nbytes, 16))
        try:
            libhip.call("ddh_memcpy_h2d", d_in, src.ctypes.data_as(C.c_void_p), src.nbytes, None)
            fn(d_in, d_out)
            libhip.call("ddh_memcpy_d2h", out.ctypes.data_as(C.c_void_p), d_out, out.nbytes, None)
            libhip.call("ddh_stream_sync", None)
        finally:
            libhip.call("ddh_free", d_in)
            libhip.call("ddh_free", d_out)
        self.dst[...] = out


def _outer_inner(shape, axis):
    """[outer][n][inner] view of an axis (the reduction FourierTransform does, fftw_wrappers.pyx:106-127)."""
    return int(math.prod(shape[:axis])), int(math.prod(shape[axis + 1:]))


class _AxisPlan:
    """forward(gdata, cdata, axis) / backward(cdata, gdata, axis) through a pair of C entry points"""
    _fwd = _bwd = None
    stream = None               # hipStream_t as c_void_p; None = default stream

    def _check(self, gdata, cdata, axis):
        if gdata.shape[axis] != self.N or cdata.shape[axis] != self.M:
            raise ValueError("array sizes along axis %d do not match the plan (%d grid, %d coefficients)"
                             % (axis, self.N, self.M))

    def forward(self, gdata, cdata, axis):
        self._check(gdata, cdata, axis)
        outer, inner = _outer_inner(tuple(gdata.shape), axis)
        _Staged(gdata, cdata).run(lambda g, c: libhip.call(self._fwd, self._forward_plan(), g, c, outer, inner, self.stream))

    def backward(self, cdata, gdata, axis):
        self._check(gdata, cdata, axis)
        outer, inner = _outer_inner(tuple(gdata.shape), axis)
        _Staged(cdata, gdata).run(lambda c, g: libhip.call(self._bwd, self._backward_plan(), c, g, outer, inner, self.stream))

    def _forward_plan(self):
        return self.plan

    def _backward_plan(self):
        return self.plan

    def __del__(self):
        for h in (getattr(self, "plan", None), getattr(self, "plan_fwd_mmt", None), getattr(self, "plan_bwd_mmt", None)):
            if h is not None and h.value:
                try:
                    libhip.call("ddh_destroy", h)
                except Exception:
                    pass


class HipRealFFT(_AxisPlan):
    """RealFourier plan: FFTWRealFFT + unpack_rescale / repack_rescale (core/transforms.py:469-565)."""
    _fwd, _bwd = "ddh_rfft_forward", "ddh_rfft_backward"

    def __init__(self, grid_size, coeff_size):
        self.N, self.M = int(grid_size), int(coeff_size)
        if self.M % 2:
            raise ValueError("RealFourier coefficient size must be even")       # core/basis.py:1097-1098
        self.KN, self.KM = (self.N - 1) // 2, (self.M - 1) // 2                  # :371-378
        self.Kmax = min(self.KN, self.KM)
        self.plan = C.c_uint64(0)
        libhip.call("ddh_plan_rfft", C.byref(self.plan), self.N, self.M)


class HipComplexFFT(_AxisPlan):
    """ComplexFourier plan: FFTWComplexFFT with resize_coeffs fused (core/transforms.py:243-330); complex128 data."""
    _fwd, _bwd = "ddh_cfft_forward", "ddh_cfft_backward"

    def __init__(self, grid_size, coeff_size):
        self.N, self.M = int(grid_size), int(coeff_size)
        self.plan = C.c_uint64(0)
        libhip.call("ddh_plan_cfft", C.byref(self.plan), self.N, self.M)


def jacobi_mmt_matrices(N, M, a, b, a0, b0, dealias_before_converting=True):
    """forward (M x N) and backward (N x M) matrices of JacobiMMT (core/transforms.py:118-158) from this package's own
    Jacobi tools (quadrature and recurrences derived in tools/jacobi.py)."""
    from .tools import jacobi
    z, w = jacobi.quadrature(N, a0, b0)
    K = max(M, N)
    base = np.zeros((K, N))
    base[:N] = jacobi.polynomials(N, a0, b0, z) * w                     # rows >= N stay zero (:131-132)
    if dealias_before_converting:
        base = base[:M]
    if (a, b) != (a0, b0):
        base = np.asarray(jacobi.conversion_matrix(base.shape[0], a0, b0, a, b) @ base)
    fwd = np.ascontiguousarray(base[:M])
    bwd = np.zeros((N, M))
    Mk = min(N, M)
    bwd[:, :Mk] = jacobi.polynomials(Mk, a, b, z).T
    return fwd, np.ascontiguousarray(bwd)


class HipJacobi(_AxisPlan):
    """Jacobi plan.  Chebyshev grids (a0 = b0 = -1/2): FFTWFastChebyshevTransform with the ultraspherical conversion as
    diagonals (core/transforms.py:801-902).  Other grids, and the forward direction with
    dealias_before_converting=False (conversion applied before truncation, :833-842), use the matrix definition
    JacobiMMT (:114-158) as a dense device GEMM (ddh_plan_mmt)."""
    _fwd, _bwd = "ddh_cheb_forward", "ddh_cheb_backward"

    def __init__(self, grid_size, coeff_size, a, b, a0, b0, dealias_before_converting=None):
        from .tools import jacobi
        self.N, self.M = int(grid_size), int(coeff_size)
        self.a, self.b, self.a0, self.b0 = a, b, a0, b0
        if dealias_before_converting is None:
            dealias_before_converting = True                        # dedalus.cfg:41
        self.dealias_before_converting = bool(dealias_before_converting)
        N, M = self.N, self.M
        converts = (a, b) != (a0, b0)
        fast = (a0 == b0 == -0.5)
        self.plan = self.plan_fwd_mmt = self.plan_bwd_mmt = None
        if fast:
            offs, bands = [], None
            if converts:
                conv = np.asarray(jacobi.conversion_matrix(M, a0, b0, a, b).todense())
                offs = [o for o in range(M) if np.any(np.abs(np.diagonal(conv, o)) > 1e-17)]
                bands = np.zeros((len(offs), M))
                for d, o in enumerate(offs):
                    bands[d, :M - o] = np.diagonal(conv, o)
            self.plan = C.c_uint64(0)
            libhip.call("ddh_plan_cheb", C.byref(self.plan), N, M, len(offs),
                        libhip.as_ip(np.ascontiguousarray(offs, dtype=np.int32)) if offs else None,
                        libhip.as_dp(np.ascontiguousarray(bands)) if offs else None)
        need_fwd_mmt = (not fast) or (converts and not self.dealias_before_converting and M < N)
        if need_fwd_mmt or not fast:
            fwd, bwd = jacobi_mmt_matrices(N, M, a, b, a0, b0, self.dealias_before_converting)
            if need_fwd_mmt:
                self.plan_fwd_mmt = C.c_uint64(0)
                libhip.call("ddh_plan_mmt", C.byref(self.plan_fwd_mmt), M, N, libhip.as_dp(fwd))
            if not fast:
                self.plan_bwd_mmt = C.c_uint64(0)
                libhip.call("ddh_plan_mmt", C.byref(self.plan_bwd_mmt), N, M, libhip.as_dp(bwd))

    def forward(self, gdata, cdata, axis):
        if self.plan_fwd_mmt is None:
            return super().forward(gdata, cdata, axis)
        self._check(gdata, cdata, axis)
        outer, inner = _outer_inner(tuple(gdata.shape), axis)
        _Staged(gdata, cdata).run(lambda g, c: libhip.call("ddh_mmt_apply", self.plan_fwd_mmt, g, c, outer, inner, self.stream))

    def backward(self, cdata, gdata, axis):
        if self.plan_bwd_mmt is None:
            return super().backward(cdata, gdata, axis)
        self._check(gdata, cdata, axis)
        outer, inner = _outer_inner(tuple(gdata.shape), axis)
        _Staged(cdata, gdata).run(lambda c, g: libhip.call("ddh_mmt_apply", self.plan_bwd_mmt, c, g, outer, inner, self.stream))


class HipSWSHColatitude:
    """SWSHColatitudeTransform (core/transforms.py:1251-1340): ctor (Ntheta, Lmax, m_maps, s); the Python loop over
    the local azimuthal wavenumbers is one grouped launch.  m_maps: the reference's list of
    (m, mg_slice, mc_slice, ell_slice) (SphereBasis.m_maps, core/basis.py:2939-2970) or the integer rows of
    curvilinear.m_maps_to_groups.  `matrices` (optional): {m: (forward [n_ell][Ntheta], backward [Ntheta][n_ell])},
    e.g. the reference's own _forward_SWSH_matrices / _backward_SWSH_matrices; default: tools/sphere.py."""
    stream = None

    def __init__(self, Ntheta, Lmax, m_maps, s, matrices=None):
        from .core.curvilinear import m_maps_to_groups
        from .tools import sphere
        self.N2g, self.N2c = int(Ntheta), int(Lmax) + 1
        self.Lmax, self.s = int(Lmax), int(s)
        groups = m_maps if isinstance(m_maps, np.ndarray) else m_maps_to_groups(m_maps, Lmax)
        self.groups = np.asarray(groups, dtype=np.int64).reshape(-1, 7)
        ms = []
        for row in self.groups:
            m = int(row[0])
            if abs(m) <= self.Lmax and m not in ms:
                ms.append(m)
        if matrices is None:
            matrices = {m: sphere.swsh_matrices(self.N2g, self.Lmax, m, self.s) for m in ms}
        index = {m: i for i, m in enumerate(ms)}
        arr = (libhip.MmtGroup * max(len(self.groups), 1))()
        for i, row in enumerate(self.groups):
            arr[i] = libhip.MmtGroup(index.get(int(row[0]), -1), *[int(v) for v in row[1:]])
        fw = [np.ascontiguousarray(matrices[m][0], dtype=np.float64) for m in ms]
        bw = [np.ascontiguousarray(matrices[m][1], dtype=np.float64) for m in ms]
        rows = np.ascontiguousarray([a.shape[0] for a in fw], dtype=np.int32)
        pf = (C.c_void_p * max(len(fw), 1))(*[a.ctypes.data for a in fw])
        pb = (C.c_void_p * max(len(bw), 1))(*[a.ctypes.data for a in bw])
        self.plan = C.c_uint64(0)
        libhip.call("ddh_plan_grouped_mmt", C.byref(self.plan), self.N2g, len(self.groups), C.cast(arr, C.c_void_p),
                    len(fw), libhip.as_ip(rows), pf, pb)

    @staticmethod
    def _reduced(shape, axis):
        """(N0, N1, N2, N3) with the colatitude axis third and the azimuthal axis second (:1296-1310)"""
        if axis < 1:
            raise ValueError("the colatitude axis follows the azimuthal axis")
        return (int(math.prod(shape[:axis - 1])), int(shape[axis - 1]), int(shape[axis]), int(math.prod(shape[axis + 1:])))

    def forward_reduced(self, gdata, cdata, gshape=None, cshape=None):
        gs = tuple(gshape or gdata.shape)
        cs = tuple(cshape or cdata.shape)
        _Staged(gdata, cdata).run(lambda g, c: libhip.call("ddh_grouped_mmt_forward", self.plan, g, c, gs[0], gs[1],
                                                           cs[1], cs[2], gs[3], self.stream))

    def backward_reduced(self, cdata, gdata, gshape=None, cshape=None):
        gs = tuple(gshape or gdata.shape)
        cs = tuple(cshape or cdata.shape)
        _Staged(cdata, gdata).run(lambda c, g: libhip.call("ddh_grouped_mmt_backward", self.plan, c, g, gs[0], gs[1],
                                                           cs[1], cs[2], gs[3], self.stream))

    def forward(self, gdata, cdata, axis):
        self.forward_reduced(gdata, cdata, self._reduced(tuple(gdata.shape), axis), self._reduced(tuple(cdata.shape), axis))

    def backward(self, cdata, gdata, axis):
        self.backward_reduced(cdata, gdata, self._reduced(tuple(gdata.shape), axis), self._reduced(tuple(cdata.shape), axis))

    def __del__(self):
        try:
            libhip.call("ddh_destroy", self.plan)
        except Exception:
            pass


class HipCommunicator:
    """RCCL communicator owned by the library (ddh_comm_*).  `bcast(bytes_or_None) -> bytes` distributes rank 0's id,
    e.g. `lambda b: mpi_comm.bcast(b, root=0)` in the reference, a torch.distributed broadcast here."""

    def __init__(self, rank, size, bcast):
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            libhip.call("ddh_comm_unique_id", buf)
        ident = bcast(bytes(buf) if rank == 0 else None)
        buf = (C.c_ubyte * 128)(*ident)
        self.rank, self.size = int(rank), int(size)
        self.handle = C.c_uint64(0)
        libhip.call("ddh_comm_create", C.byref(self.handle), self.rank, self.size, buf)

    def __del__(self):
        try:
            libhip.call("ddh_destroy", self.handle)
        except Exception:
            pass


class HipTranspose:
    """TransposePlanner drop-in (FFTWTranspose signature, core/transposes.pyx:60; selected by
    [parallelism] TRANSPOSE_LIBRARY, core/distributor.py:26-29).  `comm`: a HipCommunicator.  CL / RL are the
    column-local / row-local DEVICE arrays of the two layouts (distinct buffers)."""
    stream = None

    def __init__(self, global_shape, chunk_shape, dtype, axis, comm):
        if np.dtype(dtype) != np.float64:
            raise ValueError("float64 data only (complex fields: view as float64 with a doubled last axis)")
        gs = [int(v) for v in global_shape]
        self.global_shape, self.axis, self.comm = gs, int(axis), comm
        self.N0, self.N1, self.N2, self.N3 = math.prod(gs[:axis]), gs[axis], gs[axis + 1], math.prod(gs[axis + 2:])
        cs = [int(v) for v in chunk_shape]
        # blocks of whole chunks, as the reference's layouts deal them out (core/distributor.py): equal when the rank
        # count divides the chunk count, otherwise the last ranks own less (Alltoallv transposes, transposes.pyx:287-445)
        blk = lambda n, c: c * -(-(-(-n // c)) // comm.size)
        self.block1, self.block2 = blk(self.N1, max(cs[axis], 1)), blk(self.N2, max(cs[axis + 1], 1))
        self.plan = C.c_uint64(0)
        libhip.call("ddh_a2a_plan_blocks", C.byref(self.plan), comm.handle, self.N0, self.N1, self.N2, self.N3,
                    self.block1, self.block2)

    def _local(self, n, block):
        """Entries of an axis of length n this rank owns when it is dealt out in blocks of `block`."""
        return max(0, min(block, n - self.comm.rank * block))

    def _check(self, CL, RL):
        """The plan reads / writes exactly these many elements: a caller with another chunk convention must hear about it
        here, not through an out-of-bounds access on the device."""
        want_cl = self.N0 * self.N1 * self._local(self.N2, self.block2) * self.N3
        want_rl = self.N0 * self._local(self.N1, self.block1) * self.N2 * self.N3
        for name, arr, want in (("column-local", CL, want_cl), ("row-local", RL, want_rl)):
            have = int(arr.numel()) if hasattr(arr, "numel") else int(np.prod(arr.shape))
            if have != want:
                raise ValueError("HipTranspose: the %s array holds %d elements, the plan's block of rank %d has %d "
                                 "(global %s, axis %d, blocks %d / %d)" % (name, have, self.comm.rank, want, self.global_shape,
                                                                          self.axis, self.block1, self.block2))

    def localize_rows(self, CL, RL):
        self._check(CL, RL)
        libhip.call("ddh_a2a_localize_rows", self.plan, C.c_void_p(device_pointer(CL)), C.c_void_p(device_pointer(RL)),
                    self.stream)

    def localize_columns(self, RL, CL):
        self._check(CL, RL)
        libhip.call("ddh_a2a_localize_columns", self.plan, C.c_void_p(device_pointer(RL)),
                    C.c_void_p(device_pointer(CL)), self.stream)

    def __del__(self):
        try:
            libhip.call("ddh_destroy", self.plan)
        except Exception:
            pass


def install(register_transform, RealFourier=None, ComplexFourier=None, Jacobi=None, SphereBasis=None, name="hip"):
    """Register the plan classes with the reference's registry (core/transforms.py:27-32):
        from dedalus.core import basis, transforms
        install(transforms.register_transform, basis.RealFourier, basis.ComplexFourier, basis.Jacobi, basis.SphereBasis)
    afterwards `RealFourier(..., library='hip')` etc. select them (core/basis.py:485, 849, 2710)."""
    for cls, plan in ((RealFourier, HipRealFFT), (ComplexFourier, HipComplexFFT), (Jacobi, HipJacobi),
                      (SphereBasis, HipSWSHColatitude)):
        if cls is not None:
            register_transform(cls, name)(plan)


# ---- B2: matsolver registry (libraries/matsolvers.py:10-13) ---------------------------------------------------------
class HipBandMatsolver:
    """`cls(matrix, solver).solve(vector)` of the reference's matsolver interface (libraries/matsolvers.py:16-27;
    constructed per subproblem in core/solvers.py:95-116, core/timesteppers.py:172-181, 630-640) over the library's
    pencil engine with ONE pencil (ddh_pencil_create with nfourier = 0):

      * a matrix that is banded as given (lower bandwidth <= 16, kl + ku <= 64 -- the engine's register windows) is
        factored by the batched band LU with partial pivoting (ddh_pencil_factor) and solved by its sweeps;
      * anything else (tau columns / boundary rows far from the diagonal, as in the reference's tau-bordered pencils)
        is flagged by the engine and served by an explicit inverse applied on the device.

    Register it like any matsolver: `matsolvers.add_solver(HipBandMatsolver)`, then `matsolver = 'hipbandmatsolver'`.
    This is the fine-grained seam, kept for completeness: one launch per pencil is the loop the library exists to
    remove, the production path factors all pencils at once behind `timestepper.step`.
    Complex matrices are solved as the interleaved real system of twice the size."""

    sparse = True
    banded = False
    config = {}

    def __init__(self, matrix, solver=None):
        from scipy import sparse
        from .device import Device
        from .pencilpack import PencilPack, TermList
        A = sparse.coo_matrix(matrix)
        if A.shape[0] != A.shape[1]:
            raise ValueError("HipBandMatsolver: square matrix expected")
        self.n = int(A.shape[0])
        self.cx = bool(np.iscomplexobj(A.data))
        row, col, val = A.row.astype(np.int64), A.col.astype(np.int64), A.data
        if self.cx:
            # (x_re, x_im) interleaved per unknown: [[re, -im], [im, re]] blocks
            row = np.concatenate([2 * row, 2 * row, 2 * row + 1, 2 * row + 1])
            col = np.concatenate([2 * col, 2 * col + 1, 2 * col, 2 * col + 1])
            val = np.concatenate([val.real, -val.imag, val.imag, val.real])
        N = self.n * (2 if self.cx else 1)
        keep = val != 0
        row, col, val = row[keep], col[keep], np.asarray(val[keep], dtype=np.float64)
        kl = int(max(0, (row - col).max())) if row.size else 0
        ku = int(max(0, (col - row).max())) if row.size else 0
        self.band = (kl <= 16 and kl + ku <= 64)
        self.dev = Device.get()
        self.pack = PencilPack(self.dev, 0, N, 1, 1, np.zeros(1), np.zeros(1))
        mat = self.pack.add_matrix(TermList(N, N, row, col, val))
        perm = np.arange(N, dtype=np.int32)
        axes = np.full(N, 3, dtype=np.uint8)
        # (0 * mat + 1 * mat); a band declared as (0, 0) sends a non-banded matrix to the engine's dense path
        self.lu = self.pack.factor(mat, mat, 0.0, 1.0, perm, perm, N, kl if self.band else 0, ku if self.band else 0, axes, axes)
        self.N = N

    def solve(self, vector):
        v = np.asarray(vector)
        out_dtype = np.result_type(v.dtype, np.complex128 if self.cx else np.float64)
        cols = v.reshape(self.n, -1)
        res = np.empty(cols.shape, dtype=out_dtype)
        for j in range(cols.shape[1]):
            b = cols[:, j]
            if self.cx:
                rb = np.empty(self.N)
                rb[0::2], rb[1::2] = b.real, np.imag(b)
                parts = [rb]
            elif np.iscomplexobj(b):
                parts = [np.ascontiguousarray(b.real), np.ascontiguousarray(b.imag)]     # a real matrix, twice
            else:
                parts = [np.asarray(b, dtype=np.float64)]
            sols = []
            for p in parts:
                d_b = self.dev.from_host(np.ascontiguousarray(p).reshape(self.N, 1, 1))
                d_x = self.dev.empty((self.N, 1, 1))
                self.pack.solve(self.lu, d_b, d_x)
                self.dev.sync()
                sols.append(self.dev.to_host(d_x).reshape(self.N))
            if self.cx:
                res[:, j] = sols[0][0::2] + 1j * sols[0][1::2]
            elif len(sols) == 2:
                res[:, j] = sols[0] + 1j * sols[1]
            else:
                res[:, j] = sols[0]
        return res.reshape(v.shape)


def install_matsolver(add_solver):
    """`install_matsolver(dedalus.libraries.matsolvers.add_solver)` -> `matsolvers['hipbandmatsolver']`"""
    return add_solver(HipBandMatsolver)
