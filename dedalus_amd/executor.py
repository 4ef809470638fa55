"""
HipExecutor: the only compute back-end of the product.  Every numerical operation is a call into
libdedalus_hip.so through ctypes; torch (ROCm) provides HBM allocations, the HIP stream and, for
multi-GPU runs, torch.distributed (RCCL).  Construction fails loudly without a gfx950 device.
"""

import ctypes as C

import numpy as np

from . import libhip
from .device import Device, ptr
from .pencilpack import PencilPack


class HipExecutor:
    name = "hip"
    mmt_pairs = True            # grouped transforms accept paired right-hand sides (ddh_grouped_mmt_set_pairs)

    def __init__(self, device=None):
        self.dev = device or Device.get()
        self.torch = self.dev.torch
        self._plans = {}
        self.timer = None            # KernelTimer while bench.py measures, else None

    # ---- memory -------------------------------------------------------------------------------------
    def empty(self, shape):
        return self.dev.empty(shape)

    def zeros(self, shape):
        return self.dev.zeros(shape)

    def from_host(self, a):
        return self.dev.from_host(np.ascontiguousarray(a, dtype=np.float64))

    def download(self, t):
        return t.detach().cpu().numpy()

    def download_async(self, t):
        """Start a device-to-host copy that overlaps the following kernels: the data are snapshotted on the compute
        stream (the next step overwrites state views), then copied into pinned host memory on a side stream.
        Returns a handle whose .wait() gives the host array (analysis output staging)."""
        torch = self.torch
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream()
            self._pinned = {}
        snap = t.detach().clone()
        key = (int(snap.numel()), snap.dtype)
        pool = self._pinned.setdefault(key, [])
        host = pool.pop() if pool else torch.empty(snap.numel(), dtype=snap.dtype, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            host.copy_(snap.reshape(-1), non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        snap.record_stream(self._copy_stream)
        return _AsyncDownload(host, done, tuple(snap.shape), pool)

    def upload(self, dst, a):
        dst.copy_(self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).reshape(dst.shape))

    def copy(self, dst, src):
        if dst.data_ptr() != src.data_ptr():
            dst.reshape(-1).copy_(src.reshape(-1)) if dst.is_contiguous() else dst.copy_(src.reshape(dst.shape))

    def assign(self, dst_view, src_view):
        dst_view.copy_(src_view)

    def fill_zero(self, a):
        a.zero_()

    def sync(self):
        self.dev.sync()

    # ---- kernels -------------------------------------------------------------------------------------
    def lincomb(self, y, xs, alphas):
        if self.timer is not None:
            return self.timer.run("lincomb", (len(xs) + 1) * y.numel() * 8, self._lincomb, y, xs, alphas)
        return self._lincomb(y, xs, alphas)

    def _lincomb(self, y, xs, alphas):
        n = y.numel()
        arr = (C.c_void_p * len(xs))(*[C.c_void_p(x.data_ptr()) for x in xs])
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        libhip.call("ddh_lincomb", ptr(y), len(xs), arr, libhip.as_dp(al), n, self.dev.stream)

    def make_scatter(self, flat_idx, vals):
        """Device copy of a sparse vector (flat indices, values) for scatter_add."""
        t = self.torch
        idx = t.as_tensor(np.ascontiguousarray(flat_idx, dtype=np.int64), device=self.dev.tdev)
        return (idx, self.from_host(np.ascontiguousarray(vals, dtype=np.float64)))

    def scatter_add(self, y, sparse):
        idx, vals = sparse
        libhip.call("ddh_scatter_add", ptr(y), C.c_void_p(idx.data_ptr()), ptr(vals), idx.numel(), self.dev.stream)

    def scatter_set(self, y, sparse):
        """y[idx] = vals (a handful of entries: constants of the k = 0 pencil)."""
        idx, vals = sparse
        libhip.call("ddh_scatter_set", ptr(y), C.c_void_p(idx.data_ptr()), ptr(vals), idx.numel(), self.dev.stream)

    def bilinear(self, out, ncomp_out, a, b, npts, terms):
        if self.timer is not None:
            na = len({t[1] for t in terms}) + len({t[2] for t in terms}) + ncomp_out
            return self.timer.run("grid_bilinear", na * npts * 8, self._bilinear, out, ncomp_out, a, b, npts, terms)
        return self._bilinear(out, ncomp_out, a, b, npts, terms)

    def _bilinear(self, out, ncomp_out, a, b, npts, terms):
        ic = np.ascontiguousarray([t[0] for t in terms], dtype=np.int32)
        ia = np.ascontiguousarray([t[1] for t in terms], dtype=np.int32)
        ib = np.ascontiguousarray([t[2] for t in terms], dtype=np.int32)
        cf = np.ascontiguousarray([t[3] for t in terms], dtype=np.float64)
        libhip.call("ddh_grid_bilinear", ptr(out), ncomp_out, ptr(a), ptr(b), npts, len(terms),
                    libhip.as_ip(ic), libhip.as_ip(ia), libhip.as_ip(ib), libhip.as_dp(cf), self.dev.stream)

    FUSED_LIMITS = dict(na=3, nb=12, nc=4, terms=32, max_grid=1536)

    def fused_capable(self, spec):
        """Can the grid stage along a contiguous RealFourier axis with this plan spec run fused?"""
        return spec[0] == "rfft" and spec[1] <= self.FUSED_LIMITS["max_grid"]

    def rfft_bilinear_fused(self, spec, basis, a_list, b_list, out_list, nlines, terms, a_dscale=None,
                            b_dscale=None):
        """out[ic] = rfft_fwd(sum coef * rfft_bwd(D a[ia]) * rfft_bwd(D b[ib])) on [nlines][M] line arrays;
        a_dscale / b_dscale: per operand 2 pi / L to differentiate along the axis at load, 0 for none."""
        if self.timer is not None:
            nb = (len(a_list) + len(b_list) + len(out_list)) * nlines * spec[2] * 8
            return self.timer.run("rfft_bilinear_fused", nb, self._rfft_bilinear_fused, spec, basis, a_list, b_list,
                                  out_list, nlines, terms, a_dscale, b_dscale)
        return self._rfft_bilinear_fused(spec, basis, a_list, b_list, out_list, nlines, terms, a_dscale, b_dscale)

    def _rfft_bilinear_fused(self, spec, basis, a_list, b_list, out_list, nlines, terms, a_dscale=None,
                             b_dscale=None):
        kind, h, _ = self._plan(spec, basis)
        ads = np.ascontiguousarray(a_dscale if a_dscale is not None else np.zeros(len(a_list)), dtype=np.float64)
        bds = np.ascontiguousarray(b_dscale if b_dscale is not None else np.zeros(len(b_list)), dtype=np.float64)
        pa = (C.c_void_p * len(a_list))(*[C.c_void_p(x.data_ptr()) for x in a_list])
        pb = (C.c_void_p * len(b_list))(*[C.c_void_p(x.data_ptr()) for x in b_list])
        po = (C.c_void_p * len(out_list))(*[C.c_void_p(x.data_ptr()) for x in out_list])
        ic = np.ascontiguousarray([t[0] for t in terms], dtype=np.int32)
        ia = np.ascontiguousarray([t[1] for t in terms], dtype=np.int32)
        ib = np.ascontiguousarray([t[2] for t in terms], dtype=np.int32)
        cf = np.ascontiguousarray([t[3] for t in terms], dtype=np.float64)
        libhip.call("ddh_rfft_bilinear_fused", h, len(a_list), pa, libhip.as_dp(ads), len(b_list), pb,
                    libhip.as_dp(bds), len(out_list), po, nlines,
                    len(terms), libhip.as_ip(ic), libhip.as_ip(ia), libhip.as_ip(ib), libhip.as_dp(cf),
                    self.dev.stream)

    def _plan(self, spec, basis):
        if spec not in self._plans:
            h = C.c_uint64(0)
            kind = spec[0]
            if kind == "rfft":
                libhip.call("ddh_plan_rfft", C.byref(h), spec[1], spec[2])
                self._plans[spec] = ("rfft", h, None)
            elif kind == "cheb":
                offs, bands = basis.conversion_bands()
                if offs:
                    o = np.ascontiguousarray(offs, dtype=np.int32)
                    bnd = np.ascontiguousarray(bands)
                    libhip.call("ddh_plan_cheb", C.byref(h), spec[1], spec[2], len(offs), libhip.as_ip(o),
                                libhip.as_dp(bnd))
                else:
                    libhip.call("ddh_plan_cheb", C.byref(h), spec[1], spec[2], 0, None, None)
                self._plans[spec] = ("cheb", h, None)
            elif kind == "mmt":
                fwd, bwd = basis.mmt_matrices(spec[1])
                hf, hb = C.c_uint64(0), C.c_uint64(0)
                libhip.call("ddh_plan_mmt", C.byref(hf), fwd.shape[0], fwd.shape[1], libhip.as_dp(fwd))
                libhip.call("ddh_plan_mmt", C.byref(hb), bwd.shape[0], bwd.shape[1], libhip.as_dp(bwd))
                self._plans[spec] = ("mmt", hf, hb)
            else:
                raise NotImplementedError(kind)
        return self._plans[spec]

    def transform(self, spec, basis, direction, src, dst, outer, inner, deriv=0.0, tiled_row=0, xb=0):
        """tiled_row (strided Chebyshev only): the coefficient rows of `inner` = nx * tiled_row doubles -- dst of a forward,
        src of a backward transform -- are tile-major (ddh_cheb_forward_tiled / ddh_fft_set_coeff_tiled).  xb: the stage array of this transform (grid side of a Chebyshev, coefficient side of a
        real-Fourier transform) is x-blocked: row length ny / z planes gz (ddh_fft_set_stage_layout), 0 = natural."""
        if self.timer is not None:
            name = "%s_%s_%s" % (spec[0], direction, "strided" if inner > 1 else "contig")
            return self.timer.run(name, (src.numel() + dst.numel()) * 8, self._transform, spec, basis, direction,
                                  src, dst, outer, inner, deriv, tiled_row, xb)
        return self._transform(spec, basis, direction, src, dst, outer, inner, deriv, tiled_row, xb)

    def _stage_layout(self, h, value):
        """x-blocked stage layout of a plan (ddh_fft_set_stage_layout); the library call only when the value changes.
        value: 0 natural; an int (row length ny for a Chebyshev plan, z planes gz for a real-Fourier plan, blocks of 64
        rows); or (gz, rows) for a real-Fourier plan with blocks of `rows` = nx / P rows (ddh_fft_set_stage_block: the layout
        the all-to-all of a sharded run delivers)."""
        block, z0, cw = 0, 0, 0
        if isinstance(value, tuple):
            if len(value) == 4:                 # (gz, rows, z0, planes): a window of every component's planes
                value, block, z0, cw = value
            else:
                value, block = value
        cur = self.__dict__.setdefault("_layouts", {})
        if cur.get(h.value, (0, 0)) != (int(value), int(block)):
            libhip.call("ddh_fft_set_stage_layout", h, int(value))
            if int(block) != cur.get(h.value, (0, 0))[1]:
                libhip.call("ddh_fft_set_stage_block", h, int(block))
            cur[h.value] = (int(value), int(block))
        wcur = self.__dict__.setdefault("_windows", {})
        if wcur.get(h.value, (0, 0)) != (int(z0), int(cw)):
            libhip.call("ddh_fft_set_stage_window", h, int(z0), int(cw))
            wcur[h.value] = (int(z0), int(cw))

    # (grid, coefficient) sizes the wave-per-four-line-pairs kernels are instantiated for (csrc/ddh_fftwave.hip:
    # DDH_CHEB_WAVE_SIZES, launch_wave_rfft_kind)
    WAVE_CHEB_SIZES = frozenset([(384, 256), (192, 128), (256, 256), (192, 192), (128, 128), (64, 64), (256, 128), (128, 64)])
    WAVE_RFFT_SIZES = frozenset([(768, 512), (576, 384), (384, 256), (192, 128)])

    def stage_layout_ok(self, zspec, xspec, nx, ny):
        """Can the array between the z and the x transforms be x-blocked?  (both run as strided wave kernels)"""
        return (zspec[0] == "cheb" and (int(zspec[1]), int(zspec[2])) in self.WAVE_CHEB_SIZES and xspec[0] == "rfft"
                and (int(xspec[1]), int(xspec[2])) in self.WAVE_RFFT_SIZES and nx % 64 == 0 and ny % 8 == 0
                and int(xspec[2]) == nx)

    def stage_block_ok(self, xspec, nx, nx_local, ny):
        """Sharded run: can the x transforms read / write the exchanged layout [p][z][nx / P][ky] in place?  (real-Fourier
        wave kernels, blocks of 64 / 128 / 256 rows)"""
        return (xspec[0] == "rfft" and (int(xspec[1]), int(xspec[2])) in self.WAVE_RFFT_SIZES and int(xspec[2]) == nx
                and nx_local in (64, 128, 256) and nx % nx_local == 0 and ny % 2 == 0)

    def tiled_forward_ok(self, spec, basis, inner, row_len):
        """Can `transform(..., "forward", tiled_row=row_len)` run?  (the strided-axis wave kernel's sizes)"""
        return (spec[0] == "cheb" and (int(spec[1]), int(spec[2])) in self.WAVE_CHEB_SIZES and inner > 1
                and row_len % 8 == 0 and inner % row_len == 0 and (inner // row_len) % 8 == 0)

    def _transform(self, spec, basis, direction, src, dst, outer, inner, deriv=0.0, tiled_row=0, xb=0):
        kind, h, h2 = self._plan(spec, basis)
        if kind in ("cheb", "rfft") and inner > 1:
            self._stage_layout(h, xb)
        elif xb:
            raise NotImplementedError("x-blocked stage layout: strided Chebyshev / real-Fourier transforms only")
        if tiled_row:
            if kind != "cheb" or deriv:
                raise NotImplementedError("tile-major coefficient rows: Chebyshev transforms only")
            if direction == "forward":
                if isinstance(tiled_row, tuple):
                    raise NotImplementedError("kx-band-major rows: backward transforms of the state only")
                libhip.call("ddh_cheb_forward_tiled", h, ptr(src), ptr(dst), outer, inner, int(tiled_row), self.dev.stream)
                return
            row_len, band = tiled_row if isinstance(tiled_row, tuple) else (tiled_row, 0)
            libhip.call("ddh_fft_set_coeff_tiled", h, int(row_len), int(band))
            try:
                libhip.call("ddh_cheb_backward", h, ptr(src), ptr(dst), outer, inner, self.dev.stream)
            finally:
                libhip.call("ddh_fft_set_coeff_tiled", h, 0, 0)
            return
        if deriv:
            if kind != "rfft" or direction != "backward":
                raise NotImplementedError("derivative at load: RealFourier backward transforms only")
            libhip.call("ddh_rfft_backward_deriv", h, ptr(src), ptr(dst), outer, inner, float(deriv), self.dev.stream)
        elif kind == "mmt":
            libhip.call("ddh_mmt_apply", h if direction == "forward" else h2, ptr(src), ptr(dst), outer, inner,
                        self.dev.stream)
        else:
            libhip.call("ddh_%s_%s" % (kind, direction), h, ptr(src), ptr(dst), outer, inner, self.dev.stream)

    def transform_dual(self, spec, basis, src, dst, dst_deriv, outer, inner, dscale, xb=0):
        """dst = backward RealFourier transform of src, dst_deriv = the same of d/dx src: one read of the coefficients
        (ddh_rfft_backward_dual)."""
        if self.timer is not None:
            nb = (src.numel() + dst.numel() + dst_deriv.numel()) * 8
            return self.timer.run("rfft_backward_%s_dual" % ("strided" if inner > 1 else "contig"), nb,
                                  self._transform_dual, spec, basis, src, dst, dst_deriv, outer, inner, dscale, xb)
        return self._transform_dual(spec, basis, src, dst, dst_deriv, outer, inner, dscale, xb)

    def _transform_dual(self, spec, basis, src, dst, dst_deriv, outer, inner, dscale, xb=0):
        kind, h, _ = self._plan(spec, basis)
        if kind != "rfft":
            raise NotImplementedError("dual transform: RealFourier axes only")
        if inner > 1:
            self._stage_layout(h, xb)
        libhip.call("ddh_rfft_backward_dual", h, ptr(src), ptr(dst), ptr(dst_deriv), outer, inner, float(dscale),
                    self.dev.stream)

    def transform_dual_z(self, spec, basis, src, dst, dst_deriv, dvec, outer, inner, xb=0, tiled_row=0):
        """dst = backward Chebyshev transform of src (the family's own basis), dst_deriv = backward transform, in `basis`
        (the derivative's basis), of the one-superdiagonal operator dvec applied to src (ddh_cheb_backward_dual).
        tiled_row: the rows of src ([nx][tiled_row]) are tile-major (a state field of a solver with a tile-major state);
        (row length, rows of the state vector): kx-band-major."""
        if self.timer is not None:
            nb = (src.numel() + dst.numel() + dst_deriv.numel()) * 8
            return self.timer.run("cheb_backward_%s_dual" % ("strided" if inner > 1 else "contig"), nb,
                                  self._transform_dual_z, spec, basis, src, dst, dst_deriv, dvec, outer, inner, xb, tiled_row)
        return self._transform_dual_z(spec, basis, src, dst, dst_deriv, dvec, outer, inner, xb, tiled_row)

    def _transform_dual_z(self, spec, basis, src, dst, dst_deriv, dvec, outer, inner, xb=0, tiled_row=0):
        kind, h, _ = self._plan(spec, basis)
        if kind != "cheb":
            raise NotImplementedError("dual z transform: Chebyshev-family axes only")
        if inner > 1:
            self._stage_layout(h, xb)
        if tiled_row:
            row_len, band = tiled_row if isinstance(tiled_row, tuple) else (tiled_row, 0)
            libhip.call("ddh_fft_set_coeff_tiled", h, int(row_len), int(band))
        try:
            libhip.call("ddh_cheb_backward_dual", h, ptr(src), ptr(dst), ptr(dst_deriv), ptr(dvec), outer, inner,
                        self.dev.stream)
        finally:
            if tiled_row:
                libhip.call("ddh_fft_set_coeff_tiled", h, 0, 0)

    def tile_rows(self, src, dst, nrows, nx, ny, to_tiled, band_rows=0):
        """nrows rows of [nx][ny] doubles: natural -> tile-major (to_tiled) or back (ddh_tile_rows), out of place.
        band_rows: the tiled side is a block of rows of a kx-band-major vector of that many rows (its first row in band 0)."""
        libhip.call("ddh_tile_rows", ptr(src), ptr(dst), int(nrows), int(nx), int(ny), 1 if to_tiled else 0, int(band_rows),
                    self.dev.stream)

    def cfl_max(self, u, ncomp, shape, inv_spacings, comp_axis):
        """max over the grid of sum_c |u_c| / dx_c; inv_spacings: device arrays per component."""
        n = int(np.prod(shape))
        res = self.dev.zeros((1,))
        arr = (C.c_void_p * ncomp)(*[C.c_void_p(a.data_ptr()) for a in inv_spacings])
        ca = np.ascontiguousarray(comp_axis, dtype=np.int32)
        ln = (C.c_long * len(shape))(*[int(x) for x in shape])
        libhip.call("ddh_grid_cfl", ptr(res), ptr(u), ncomp, n, arr, libhip.as_ip(ca), ln, len(shape), self.dev.stream)
        return float(res.cpu().item())

    def cfl_max_spherical(self, u, inv_h, inv_dr):
        """u [3][Nphi][Ntheta][Nr] (device), inv_h / inv_dr device arrays over r -> max CFL frequency (float)."""
        nr = int(u.shape[-1])
        n_ang = int(u.shape[1]) * int(u.shape[2])
        res = self.empty((1,))
        libhip.call("ddh_grid_cfl_spherical", ptr(res), ptr(u), n_ang, nr, ptr(inv_h), ptr(inv_dr), self.dev.stream)
        return float(self.download(res)[0])

    def reduce3(self, x):
        """(min, max, sum) of a device array: one reduction launch pair + a 24-byte copy to the host (ddh_grid_reduce)."""
        if not hasattr(self, "_red_work"):
            self._red_work = self.dev.empty((3072,))
            self._red_out = self.dev.empty((3,))
        x = x if x.is_contiguous() else x.contiguous()
        libhip.call("ddh_grid_reduce", ptr(self._red_out), ptr(x), int(x.numel()), ptr(self._red_work), self.dev.stream)
        mn, mx, sm = self._red_out.cpu().tolist()
        return mn, mx, sm

    def a2a_plan(self, pcomm, n0, n1, n2, n3):
        """Library-owned transpose plan (ddh_a2a_plan on the RCCL communicator of `pcomm`), cached per shape; None when
        the exchange goes through torch.distributed (parallel.Comm.library_comm)."""
        comm = pcomm.library_comm()
        if comm is None:
            return None
        key = ("a2a", int(n0), int(n1), int(n2), int(n3))
        if key not in self._plans:
            h = C.c_uint64(0)
            libhip.call("ddh_a2a_plan", C.byref(h), comm, int(n0), int(n1), int(n2), int(n3))
            self._plans[key] = h
        return self._plans[key]

    def a2a_localize_rows(self, plan, cl, rl):
        """CL [n0][n1][n2/P][n3] -> RL [n0][n1/P][n2][n3]: gathers axis 2, splits axis 1"""
        libhip.call("ddh_a2a_localize_rows", plan, ptr(cl), ptr(rl), self.dev.stream)

    def a2a_localize_columns(self, plan, rl, cl):
        """RL [n0][n1/P][n2][n3] -> CL [n0][n1][n2/P][n3]: gathers axis 1, splits axis 2"""
        libhip.call("ddh_a2a_localize_columns", plan, ptr(rl), ptr(cl), self.dev.stream)

    def a2a_pack(self, src, dst, outer, na, nb, inner, P):
        if self.timer is not None:           # (bytes: one read + one write of the packed buffer)
            return self.timer.run("a2a_pack", 2 * dst.numel() * 8, libhip.call, "ddh_a2a_pack", ptr(src), ptr(dst), outer, na, nb,
                                  inner, P, self.dev.stream)
        libhip.call("ddh_a2a_pack", ptr(src), ptr(dst), outer, na, nb, inner, P, self.dev.stream)

    def a2a_unpack(self, src, dst, outer, na, nb, inner, P):
        if self.timer is not None:
            return self.timer.run("a2a_unpack", 2 * src.numel() * 8, libhip.call, "ddh_a2a_unpack", ptr(src), ptr(dst), outer, na,
                                  nb, inner, P, self.dev.stream)
        libhip.call("ddh_a2a_unpack", ptr(src), ptr(dst), outer, na, nb, inner, P, self.dev.stream)

    def make_recombination(self, slot_map, mats):
        """Device copy of (slot -> matrix index map [n1][n2] int32, matrices [nmats][nc][nc])."""
        t = self.torch
        sm = t.as_tensor(np.ascontiguousarray(slot_map, dtype=np.int32), device=self.dev.tdev)
        return (sm, self.from_host(np.ascontiguousarray(mats, dtype=np.float64)), int(len(mats)))

    def regularity_recombine(self, data, table, radial_factor_d=None):
        """In place on data[ncomp][n1][n2][n3] (device)."""
        nc, n1, n2, n3 = [int(x) for x in data.shape]
        sm, mats, nm = table if table is not None else (None, None, 0)
        libhip.call("ddh_regularity_recombine", ptr(data), nc, n1, n2, n3,
                    C.c_void_p(sm.data_ptr()) if sm is not None else None, nm, ptr(mats) if mats is not None else None,
                    ptr(radial_factor_d) if radial_factor_d is not None else None, self.dev.stream)

    def make_grouped_mmt(self, n_grid, groups, ms, fwd_mats, bwd_mats):
        return GroupedMmt(self, n_grid, groups, ms, fwd_mats, bwd_mats)

    # ---- sphere coefficient-space kernels (csrc/ddh_sphere.hip) -----------------------------------------------
    def spin_recombine(self, src, dst, mat):
        """dst[(c', p')] = sum mat[(c', p'), (c, p)] src[(c, p)] on [ncomp][2 m + p][inner] arrays."""
        nc, n2, inner = [int(x) for x in src.shape]
        m = np.ascontiguousarray(mat, dtype=np.float64)
        libhip.call("ddh_spin_recombine", ptr(src), ptr(dst), nc, n2 // 2, inner, libhip.as_dp(m), self.dev.stream)

    def make_sphere_terms(self, nm, nl, ncomp_out, terms):
        """terms: list of (co, ci, d, coef complex [nm][nl]) -> device term list (ddh_sphere_terms_create)."""
        return SphereTerms(self, nm, nl, ncomp_out, terms)

    def make_ell_terms(self, nm, nl, nr, ncomp_out, terms, slot_map=None):
        """terms: list of (co, ci, mats [nmat][nr][nr]); slot_map [2 nm][nl] -> matrix index or -1 (default: ell where
        ell >= m) -> device radial operator (ddh_ell_terms_create)."""
        return EllTerms(self, nm, nl, nr, ncomp_out, terms, slot_map)

    def make_cgemv_batch(self, nm, nl, ncomp, mats):
        """mats: per m a complex (n_m, n_m) array, n_m = ncomp * max(nl - m, 0)."""
        return CgemvBatch(self, nm, nl, ncomp, mats)

    # ---- device factorization of the curvilinear subproblems (csrc/ddh_dense.hip) ------------------------------------
    def make_dense_inverse(self, Ms, Ls, row_valid, col_valid, complex_=False):
        """Ms / Ls: per system the dense matrices of M and L; row_valid / col_valid: per system boolean masks.
        -> object with compute(a, b): concatenated (a M + b L)^-1 on the valid blocks, on the device."""
        return DenseInverse(self, Ms, Ls, row_valid, col_valid, complex_)

    def make_ell_band(self, plan, ncomp, nslots, nl, nr, slot_limit, offsets=None):
        """Band LU of per-ell systems laid out by core/ellband.py::EllBandPlan (system vectors [ncomp][nslots][nl][nr];
        slot_limit[g]: leading slots that can hold modes of group g)."""
        return EllBand(self, plan, ncomp, nslots, nl, nr, slot_limit, offsets=offsets)

    def from_host_int64(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64), device=self.dev.tdev)

    def gather_complex_inverse(self, x, off, count, ncomp, nl, nm, nslots):
        """ddh_ellband_gather_complex_inverse: per-m complex inverses (count complex numbers, offsets off[m]) from the unit
        solves x [2 ncomp][nslots][nm][nl] of the real-form transposed systems"""
        if getattr(self, "_cinv_out", None) is None or self._cinv_out.numel() != 2 * count:
            self._cinv_out = self.dev.empty((2 * max(count, 1),))
        libhip.call("ddh_ellband_gather_complex_inverse", ptr(x), ptr(self._cinv_out), ptr(off), int(ncomp), int(nl), int(nm),
                    int(nslots), self.dev.stream)
        return self._cinv_out

    def dense_group_solve(self, inv, rhs4, x4, g):
        """x4[:, :, g, :] = inv @ rhs4[:, :, g, :] for one group kept on the dense path (inv: (ncomp nr)^2, device)"""
        R, S, _, nr = rhs4.shape
        v = rhs4[:, :, g, :].permute(0, 2, 1).reshape(R * nr, S)
        x4[:, :, g, :] = (inv.reshape(R * nr, R * nr) @ v).reshape(R, nr, S).permute(0, 2, 1)

    def make_cgemv_batch_flat(self, nm, nl, ncomp, flat, old=None):
        """per-m complex matrices given as one concatenated device array (DenseInverse.compute); old: a batch of the same
        shape to refill (a change of the timestep then allocates and frees nothing)"""
        same = isinstance(old, CgemvBatch) and getattr(old, "shape", None) == (nm, nl, ncomp)
        batch = old if same else CgemvBatch(self, nm, nl, ncomp, None)
        batch.shape = (nm, nl, ncomp)
        dst = C.c_void_p()
        libhip.call("ddh_cgemv_batch_mats", batch.handle, C.byref(dst))
        libhip.call("ddh_memcpy_d2d", dst, ptr(flat), int(flat.numel()) * 8, self.dev.stream)
        batch.nbytes = int(flat.numel()) * 8
        return batch

    def make_ell_terms_from_dense(self, nm, nl, nr, ncomp, flat, old=None):
        """Term list of the per-ell dense blocks of `flat` = [nl][ncomp nr][ncomp nr] (device): the per-ell LHS inverses.
        Sizes that tile the FP64 MFMA GEMM stay on the device end to end (old: a handle of the same shape to refill)."""
        if nr % 64 == 0 and (ncomp * nr) % 64 == 0:
            terms = old if isinstance(old, DenseEllTerms) else DenseEllTerms(self, nm, nl, nr, ncomp)
            terms.fill(flat)
            return terms
        inv = self.download(flat).reshape(nl, ncomp * nr, ncomp * nr)
        blocks = []
        for co in range(ncomp):
            for ci in range(ncomp):
                blk = inv[:, co * nr:(co + 1) * nr, ci * nr:(ci + 1) * nr]
                if np.any(blk != 0):
                    blocks.append((co, ci, np.ascontiguousarray(blk)))
        return self.make_ell_terms(nm, nl, nr, ncomp, blocks)

    def make_pack(self, nf, nrows, nx, ny, kx, ky, mx_offset=0):
        pk = PencilPack(self.dev, nf, nrows, nx, ny, kx, ky, mx_offset)
        pk.executor = self
        return pk


class _AsyncDownload:
    def __init__(self, host, done, shape, pool):
        self.host, self.done, self.shape, self.pool = host, done, shape, pool

    def wait(self):
        """host copy of the snapshot (a fresh array; the pinned buffer goes back to the pool)"""
        self.done.synchronize()
        out = self.host.numpy().reshape(self.shape).copy()
        self.pool.append(self.host)
        self.host = None
        return out


class GroupedMmt:
    """ddh_plan_grouped_mmt handle: groups = rows (m, g_start, c_start, count, ell_start, ell_step, n_ell),
    ms = the distinct m that own a matrix pair (fwd [n_ell][n_grid], bwd [n_grid][n_ell])."""

    def __init__(self, ex, n_grid, groups, ms, fwd_mats, bwd_mats):
        self.ex, self.n_grid = ex, int(n_grid)
        index = {int(m): i for i, m in enumerate(ms)}
        arr = (libhip.MmtGroup * max(len(groups), 1))()
        for i, row in enumerate(groups):
            m = int(row[0])
            arr[i] = libhip.MmtGroup(index.get(m, -1), int(row[1]), int(row[2]), int(row[3]), int(row[4]), int(row[5]),
                                     int(row[6]))
        fw = [np.ascontiguousarray(a, dtype=np.float64) for a in fwd_mats]
        bw = [np.ascontiguousarray(a, dtype=np.float64) for a in bwd_mats]
        rows = np.ascontiguousarray([a.shape[0] for a in fw], dtype=np.int32)
        pf = (C.c_void_p * max(len(fw), 1))(*[a.ctypes.data for a in fw])
        pb = (C.c_void_p * max(len(bw), 1))(*[a.ctypes.data for a in bw])
        self.handle = C.c_uint64(0)
        libhip.call("ddh_plan_grouped_mmt", C.byref(self.handle), self.n_grid, len(groups), C.cast(arr, C.c_void_p),
                    len(fw), libhip.as_ip(rows), pf, pb)
        # algorithmic work of one application (libhip.note_cost): every group multiplies its [n_ell][n_grid] matrix with
        # `count` right-hand-side slices of n3 columns; every distinct matrix is streamed once
        self._madds = sum(int(r[6]) * self.n_grid * int(r[3]) for r in groups if index.get(int(r[0]), -1) >= 0)
        self._mat_bytes = sum(a.nbytes for a in fw)
        self._pairs = 0

    def set_pairs(self, pair_g, pair_c, pair_mode, parity):
        """second right-hand-side set per group served by the same matrices (ddh_grouped_mmt_set_pairs)"""
        a = [np.ascontiguousarray(v, dtype=np.int32) for v in (pair_g, pair_c, pair_mode, parity)]
        libhip.call("ddh_grouped_mmt_set_pairs", self.handle, len(a[0]), *[libhip.as_ip(v) for v in a])
        self._pairs = int(np.count_nonzero(np.asarray(pair_g) >= 0))

    def _cost(self, name, g, c):
        n0, n3 = int(g.shape[0]), int(g.shape[3])
        madds = self._madds * (2 if self._pairs else 1) * n0 * n3          # (a paired group serves two right-hand-side sets)
        libhip.note_cost(name, 2.0 * madds, self._mat_bytes + (g.numel() + c.numel()) * 8)

    def _dims(self, g, c):
        n0, n1g, nt, n3 = [int(x) for x in g.shape]
        c0, n1c, n2c, c3 = [int(x) for x in c.shape]
        if nt != self.n_grid or c0 != n0 or c3 != n3:
            raise ValueError("grouped transform: inconsistent reduced shapes %s / %s" % (tuple(g.shape), tuple(c.shape)))
        return n0, n1g, n1c, n2c, n3

    def forward(self, g, c):
        self._cost("ddh_grouped_mmt_forward", g, c)
        libhip.call("ddh_grouped_mmt_forward", self.handle, ptr(g), ptr(c), *self._dims(g, c), self.ex.dev.stream)

    def backward(self, c, g):
        self._cost("ddh_grouped_mmt_backward", g, c)
        libhip.call("ddh_grouped_mmt_backward", self.handle, ptr(c), ptr(g), *self._dims(g, c), self.ex.dev.stream)


class SphereTerms:
    def __init__(self, ex, nm, nl, ncomp_out, terms):
        self.ex = ex
        terms = sorted(terms, key=lambda t: t[0])
        co = np.ascontiguousarray([t[0] for t in terms], dtype=np.int32)
        ci = np.ascontiguousarray([t[1] for t in terms], dtype=np.int32)
        d = np.ascontiguousarray([t[2] for t in terms], dtype=np.int32)
        coef = np.zeros((max(len(terms), 1), nm, nl), dtype=np.complex128)
        for i, t in enumerate(terms):
            coef[i] = t[3]
        self.handle = C.c_uint64(0)
        libhip.call("ddh_sphere_terms_create", C.byref(self.handle), int(nm), int(nl), int(ncomp_out), len(terms),
                    libhip.as_ip(co), libhip.as_ip(ci), libhip.as_ip(d), libhip.as_dp(coef.view(np.float64)))

    def apply(self, x, y):
        libhip.call("ddh_sphere_terms_apply", self.handle, ptr(x), ptr(y), self.ex.dev.stream)

    def __del__(self):
        try:
            libhip.call("ddh_destroy", self.handle)
        except Exception:
            pass


def default_slot_map(nm, nl):
    i1, ell = np.indices((2 * nm, nl))
    return np.where(i1 // 2 <= ell, ell, -1).astype(np.int32)


class EllTerms:
    """Device term list.  With the default slot map, blocks that are mostly full (the per-ell LHS inverses between shell
    variables) go to a second handle that runs as an FP64 MFMA GEMM and accumulates onto the banded part."""

    def __init__(self, ex, nm, nl, nr, ncomp_out, terms, slot_map=None, _split=True):
        self.ex = ex
        self.dense_part = None
        if _split and slot_map is None and nr % 64 == 0 and (ncomp_out * nr) % 64 == 0 and len(terms) > 1:
            fill = [np.count_nonzero(t[2]) / t[2].size for t in terms]
            dense = [t for t, f in zip(terms, fill) if f > 0.5]
            sparse = [t for t, f in zip(terms, fill) if f <= 0.5]
            if dense and sparse:
                self.dense_part = EllTerms(ex, nm, nl, nr, ncomp_out, dense, None, _split=False)
                terms = sparse
        terms = sorted(terms, key=lambda t: t[0])
        co = np.ascontiguousarray([t[0] for t in terms], dtype=np.int32)
        ci = np.ascontiguousarray([t[1] for t in terms], dtype=np.int32)
        nmat = int(terms[0][2].shape[0]) if terms else nl
        mats = np.zeros((max(len(terms), 1), nmat, nr, nr))
        for i, t in enumerate(terms):
            mats[i] = t[2]
        sm = np.ascontiguousarray(default_slot_map(nm, nl) if slot_map is None else slot_map, dtype=np.int32)
        self.handle = C.c_uint64(0)
        libhip.call("ddh_ell_terms_create", C.byref(self.handle), int(nm), int(nl), int(nr), int(ncomp_out), len(terms),
                    libhip.as_ip(co), libhip.as_ip(ci), nmat, libhip.as_dp(mats), libhip.as_ip(sm))
        # algorithmic work of one application: every non-zero of A[mat] times the slots that use that matrix
        uses = np.bincount(sm[sm >= 0].ravel(), minlength=nmat)[:nmat] if nmat else np.zeros(0)
        nnz = np.count_nonzero(mats.reshape(mats.shape[0], nmat, -1), axis=2).sum(axis=0) if terms else np.zeros(nmat)
        self._madds = float(np.dot(nnz, uses))
        self._mat_bytes = float(np.count_nonzero(mats)) * 8

    def apply(self, x, y):
        libhip.note_cost("ddh_ell_terms_apply", 2.0 * self._madds, self._mat_bytes + (x.numel() + y.numel()) * 8)
        libhip.call("ddh_ell_terms_apply", self.handle, ptr(x), ptr(y), self.ex.dev.stream)
        if self.dense_part is not None:
            libhip.note_cost("ddh_ell_terms_apply_acc", 2.0 * self.dense_part._madds, self.dense_part._mat_bytes + (x.numel() + y.numel()) * 8)
            libhip.call("ddh_ell_terms_apply_acc", self.dense_part.handle, ptr(x), ptr(y), 1, self.ex.dev.stream)

    def __del__(self):
        try:
            libhip.call("ddh_destroy", self.handle)
        except Exception:
            pass


class DenseInverse:
    """ddh_dense_inverse_*: M and L of every subproblem live on the device; compute(a, b) forms and inverts a M + b L
    there (no host linear algebra, no host round trip when the timestep changes)."""

    def __init__(self, ex, Ms, Ls, row_valid, col_valid, complex_):
        self.ex, self.cx = ex, bool(complex_)
        dt = np.complex128 if self.cx else np.float64
        n = np.ascontiguousarray([a.shape[0] for a in Ms], dtype=np.int32)
        cat = lambda mats: np.concatenate([np.ascontiguousarray(a, dtype=dt).ravel() for a in mats] + [np.zeros(0, dtype=dt)])
        M, L = cat(Ms), cat(Ls)
        rv = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.uint8).ravel() for v in row_valid] + [np.zeros(0, np.uint8)]))
        cv = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.uint8).ravel() for v in col_valid] + [np.zeros(0, np.uint8)]))
        self.handle = C.c_uint64(0)
        libhip.call("ddh_dense_inverse_create", C.byref(self.handle), len(n), libhip.as_ip(n), int(self.cx),
                    libhip.as_dp(M.view(np.float64)) if M.size else None, libhip.as_dp(L.view(np.float64)) if L.size else None,
                    libhip.as_ubp(rv), libhip.as_ubp(cv))
        cnt = C.c_long(0)
        libhip.call("ddh_dense_inverse_elements", self.handle, C.byref(cnt))
        self.count = cnt.value
        self._out = None

    def compute(self, a, b):
        if self._out is None:
            self._out = self.ex.dev.empty((max(self.count, 1),))
        bad = C.c_int(0)
        libhip.call("ddh_dense_inverse_compute", self.handle, float(a), float(b), ptr(self._out), C.byref(bad),
                    self.ex.dev.stream)
        if bad.value:
            raise libhip.DdhError("%d subproblem matrices are singular (a = %g, b = %g)" % (bad.value, a, b))
        return self._out

    def __del__(self):
        try:
            libhip.call("ddh_destroy", self.handle)
        except Exception:
            pass


class EllBand:
    """ddh_ellband_*: device band LU + sweeps of the per-ell systems (csrc/ddh_ellband.hip)."""

    def __init__(self, ex, plan, ncomp, nslots, nl, nr, slot_limit, offsets=None):
        """offsets = (rowoff [nl][nmax], coloff [nl][nmax], slot_stride): element offsets of the permuted rows / columns
        inside one slot and the distance between slots, for system vectors that are not [component][slot][group][n]."""
        self.ex, self.plan = ex, plan
        self.shape = (int(ncomp), int(nslots), int(nl), int(nr))
        comp_stride = nslots * nl * nr
        g = np.arange(nl)[:, None]

        def _offsets(idx):
            off = (idx // nr) * comp_stride + g * nr + idx % nr
            return np.ascontiguousarray(np.where(idx >= 0, off, 0), dtype=np.int64)
        if offsets is None:
            rowoff, coloff, slot_stride = _offsets(plan.row_index), _offsets(plan.col_index), nl * nr
        else:
            rowoff, coloff = (np.ascontiguousarray(a, dtype=np.int64).reshape(nl, plan.nmax) for a in offsets[:2])
            slot_stride = int(offsets[2])
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        self.handle = C.c_uint64(0)
        lp = lambda a: a.ctypes.data_as(C.POINTER(C.c_long))
        libhip.call("ddh_ellband_create", C.byref(self.handle), int(nl), int(plan.nmax), int(plan.kl), int(plan.ku),
                    int(plan.mp), int(plan.nbc), int(nslots), int(slot_stride), libhip.as_ip(c(plan.n, np.int32)),
                    libhip.as_ip(c(plan.nbc_of, np.int32)), libhip.as_ip(c(slot_limit, np.int32)), lp(rowoff), lp(coloff),
                    libhip.as_dp(c(plan.T, np.float64)), libhip.as_dp(c(plan.P, np.float64)),
                    libhip.as_dp(c(plan.MB, np.float64)), libhip.as_dp(c(plan.LB, np.float64)))
        self.count = 0
        # algorithmic cost of one solve (every (group, slot) column that can hold modes)
        cols = float(np.dot(plan.n.astype(np.float64), np.minimum(np.asarray(slot_limit), nslots)))
        self._flops = 2.0 * (2 * plan.kl + plan.ku + plan.mp) * cols
        self._bytes = 16.0 * cols + float(plan.n.sum()) * 8 * (2 * plan.kl + plan.ku + 1 + plan.mp)

    def factor(self, a, b, index=None):
        """-> index of the factorization (index: one to overwrite)"""
        if index is None:
            index = self.count
        bad = C.c_int(0)
        libhip.call("ddh_ellband_factor", self.handle, int(index), float(a), float(b), C.byref(bad), self.ex.dev.stream)
        if bad.value:
            raise libhip.DdhError("%d zero pivots in the band LU (a = %g, b = %g)" % (bad.value, a, b))
        self.count = max(self.count, index + 1)
        return index

    def solve(self, index, rhs, x):
        libhip.note_cost("ddh_ellband_solve", self._flops, self._bytes)
        libhip.call("ddh_ellband_solve", self.handle, int(index), ptr(rhs), ptr(x), self.ex.dev.stream)

    def info(self):
        nw, wt, nb = C.c_int(0), C.c_int(0), C.c_long(0)
        libhip.call("ddh_ellband_info", self.handle, C.byref(nw), C.byref(wt), C.byref(nb))
        return dict(nw=nw.value, wt=wt.value, factor_bytes=nb.value)

    def __del__(self):
        try:
            libhip.call("ddh_destroy", self.handle)
        except Exception:
            pass


class BorderedBandInverse:
    """Inverse of a real bordered pencil [[B, c], [r^T, e]] whose n x n band block B is singular only through one vanishing
    column j0 (the k = 0 pencil of a Cartesian problem with a pressure gauge), formed on the device at the cost of a band
    factorization and n unit solves instead of an O(n^3) dense inversion: with columns j0 and n exchanged the matrix is
    block triangular around B2 = B with column j0 replaced by c -- a band again -- so that A^-1 follows from X = B2^-1 (the
    band LU + sweeps of csrc/ddh_ellband.hip on an n x n identity, unit solve s in lane s) and one weighted column sum
    (ddh_ellband_bordered_inverse).  Replaces the reference's sparse LU of that subproblem at every change of the timestep
    (core/timesteppers.py:630-640, libraries/matsolvers.py:126-149)."""

    def __init__(self, ex, Md, Ld, n, j0):
        from .core.ellband import BandBlockPlan
        self.ex, self.n, self.j0 = ex, int(n), int(j0)
        N = n + 1
        M2, L2 = np.array(Md[:n, :n], dtype=np.float64), np.array(Ld[:n, :n], dtype=np.float64)
        M2[:, j0], L2[:, j0] = Md[:n, n], Ld[:n, n]
        plan = BandBlockPlan(M2, L2)
        self.kl, self.ku = plan.kl, plan.ku
        # system vectors [row][slot]: slot = right-hand side s in the lanes (stride 1), row / column i at i * nslots
        off = np.arange(n, dtype=np.int64)[None, :] * n
        self.band = EllBand(ex, plan, 1, n, 1, n, [n], offsets=(off, off, 1))
        self.rhs = ex.from_host(np.eye(n))
        self.x = ex.zeros((n, n))
        wM, wL = np.array(Md[n, :n], dtype=np.float64), np.array(Ld[n, :n], dtype=np.float64)
        self.dM, self.dL = float(wM[j0]), float(wL[j0])
        wM[j0], wL[j0] = Md[n, n], Ld[n, n]
        self.wM, self.wL = ex.from_host(wM), ex.from_host(wL)
        self.out = ex.zeros((N, N))

    def compute(self, a, b):
        """-> device array (n + 1)^2, the inverse of a M + b L in the permuted order; None after a zero pivot"""
        try:
            self.band.factor(a, b, index=0)
        except libhip.DdhError:
            return None
        self.band.solve(0, self.rhs, self.x)
        libhip.call("ddh_ellband_bordered_inverse", ptr(self.x), self.n, self.j0, ptr(self.wM), ptr(self.wL), self.dM, self.dL,
                    float(a), float(b), ptr(self.out), self.ex.dev.stream)
        return self.out


class DenseEllTerms:
    """All ncomp x ncomp dense blocks per ell (the shell's LHS inverses) applied as FP64 MFMA GEMMs; filled on the
    device from DenseInverse.compute (ddh_ell_blocks_from_dense), blocks that vanish for every ell are skipped."""
    dense_part = None

    def __init__(self, ex, nm, nl, nr, ncomp):
        self.ex, self.nl, self.nr, self.ncomp = ex, int(nl), int(nr), int(ncomp)
        self.handle = C.c_uint64(0)
        libhip.call("ddh_ell_terms_create_dense", C.byref(self.handle), int(nm), int(nl), int(nr), int(ncomp))

    def fill(self, flat):
        dst = C.c_void_p()
        libhip.call("ddh_ell_terms_mats", self.handle, C.byref(dst))
        libhip.call("ddh_ell_blocks_from_dense", ptr(flat), dst, self.nl, self.ncomp, self.nr, self.ex.dev.stream)
        libhip.call("ddh_ell_terms_prune", self.handle, self.ex.dev.stream)
        self._blocks = None
        if libhip.cost_log is not None:          # (measurement runs only) blocks that survive the pruning
            nz = (flat.reshape(self.nl, self.ncomp, self.nr, self.ncomp, self.nr) != 0).any(dim=4).any(dim=2).any(dim=0)
            self._blocks = int(nz.sum().item())

    def apply(self, x, y):
        if libhip.cost_log is not None:
            blocks = self._blocks if self._blocks is not None else self.ncomp * self.ncomp
            slots = sum(2 * (min(l, x.shape[1] // 2 - 1) + 1) for l in range(self.nl))      # (m, part) slots with m <= ell
            libhip.note_cost("ddh_ell_terms_apply", 2.0 * blocks * self.nr * self.nr * slots,
                             blocks * self.nl * self.nr * self.nr * 8 + (x.numel() + y.numel()) * 8)
        libhip.call("ddh_ell_terms_apply", self.handle, ptr(x), ptr(y), self.ex.dev.stream)

    def __del__(self):
        try:
            libhip.call("ddh_destroy", self.handle)
        except Exception:
            pass


class CgemvBatch:
    def __init__(self, ex, nm, nl, ncomp, mats):
        self.ex = ex
        self.handle = C.c_uint64(0)
        if mats is None:                    # storage only: filled on the device (make_cgemv_batch_flat)
            self.nbytes = 0
            libhip.call("ddh_cgemv_batch_create", C.byref(self.handle), int(nm), int(nl), int(ncomp), None)
            return
        flat = np.concatenate([np.ascontiguousarray(a, dtype=np.complex128).ravel() for a in mats] +
                              [np.zeros(0, dtype=np.complex128)])
        self.nbytes = flat.nbytes
        libhip.call("ddh_cgemv_batch_create", C.byref(self.handle), int(nm), int(nl), int(ncomp),
                    libhip.as_dp(flat.view(np.float64)) if flat.size else None)

    def apply(self, x, y):
        libhip.note_cost("ddh_cgemv_batch_apply", 8.0 * self.nbytes / 16, self.nbytes + (x.numel() + y.numel()) * 8)
        libhip.call("ddh_cgemv_batch_apply", self.handle, ptr(x), ptr(y), self.ex.dev.stream)

    def __del__(self):
        try:
            libhip.call("ddh_destroy", self.handle)
        except Exception:
            pass


class KernelTimer:
    """HIP-event timing of every kernel family on the launch stream (bench.py roofline numbers)."""

    def __init__(self, torch):
        self.torch = torch
        self.records = {}            # name -> [bytes_total, [(e0, e1), ...]]

    def run(self, name, nbytes, fn, *args):
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args)
        e1.record()
        rec = self.records.setdefault(name, [0, []])
        rec[0] += nbytes
        rec[1].append((e0, e1))
        return out

    def summary(self):
        self.torch.cuda.synchronize()
        out = {}
        for name, (nbytes, evs) in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b in evs)
            out[name] = dict(launches=len(evs), total_ms=ms, avg_ms=ms / len(evs), bytes_per_launch=nbytes / len(evs),
                             gbps=(nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0)
        return out
