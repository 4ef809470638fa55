"""
ctypes binding of libdedalus_hip.so (the C ABI declared in include/dedalus_hip.h).

The product path has no CPU fallback: if the shared library is missing, or no gfx950 device is
visible when a device operation is requested, this module raises.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DDH_LIB: another build of the same library, for A/B timing of kernel variants)
LIB_PATH = os.environ.get("DDH_LIB") or os.path.join(_HERE, "csrc", "libdedalus_hip.so")

_lib = None


class DdhError(RuntimeError):
    pass


class PencilGeom(C.Structure):
    _fields_ = [("nfourier", C.c_int), ("nrows", C.c_int), ("nx", C.c_long), ("ny", C.c_long),
                ("kx_h", C.POINTER(C.c_double)), ("ky_h", C.POINTER(C.c_double)), ("mx_offset", C.c_long)]


class PolyMat(C.Structure):
    _fields_ = [("nterms", C.c_int),
                ("row_h", C.POINTER(C.c_int)), ("col_h", C.POINTER(C.c_int)),
                ("coef_re_h", C.POINTER(C.c_double)), ("coef_im_h", C.POINTER(C.c_double)),
                ("ex_h", C.POINTER(C.c_byte)), ("ey_h", C.POINTER(C.c_byte)),
                ("dx_h", C.POINTER(C.c_byte)), ("dy_h", C.POINTER(C.c_byte))]


_vp, _h, _hp = C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)
_dp, _ip, _l, _i, _d = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_long, C.c_int, C.c_double

# name -> argtypes; every function returns int status (except ddh_last_error)
SIGNATURES = {
    "ddh_init": [_i],
    "ddh_device_count": [_ip],
    "ddh_alloc": [C.POINTER(_vp), C.c_size_t],
    "ddh_free": [_vp],
    "ddh_memset": [_vp, _i, C.c_size_t, _vp],
    "ddh_memcpy_h2d": [_vp, _vp, C.c_size_t, _vp],
    "ddh_memcpy_d2h": [_vp, _vp, C.c_size_t, _vp],
    "ddh_memcpy_d2d": [_vp, _vp, C.c_size_t, _vp],
    "ddh_stream_sync": [_vp],
    "ddh_destroy": [_h],
    "ddh_plan_rfft": [_hp, _i, _i],
    "ddh_scatter_add": [_vp, _vp, _vp, _l, _vp],
    "ddh_scatter_set": [_vp, _vp, _vp, _l, _vp],
    "ddh_pencil_set_state_tiled": [_h, _i],
    "ddh_fft_set_coeff_tiled": [_h, _l, _l],
    "ddh_tile_rows": [_vp, _vp, _l, _l, _l, _i, _l, _vp],
    "ddh_plan_grouped_mmt": [_hp, _i, _i, _vp, _i, _ip, C.POINTER(_vp), C.POINTER(_vp)],
    "ddh_grouped_mmt_set_pairs": [_h, _i, _ip, _ip, _ip, _ip],
    "ddh_grouped_mmt_forward": [_h, _vp, _vp, _l, _l, _l, _l, _l, _vp],
    "ddh_grouped_mmt_backward": [_h, _vp, _vp, _l, _l, _l, _l, _l, _vp],
    "ddh_regularity_recombine": [_vp, _i, _l, _l, _l, _vp, _i, _vp, _vp, _vp],
    "ddh_spin_recombine": [_vp, _vp, _i, _l, _l, _dp, _vp],
    "ddh_sphere_terms_create": [_hp, _i, _i, _i, _i, _ip, _ip, _ip, _dp],
    "ddh_sphere_terms_apply": [_h, _vp, _vp, _vp],
    "ddh_cgemv_batch_create": [_hp, _i, _i, _i, _dp],
    "ddh_cgemv_batch_apply": [_h, _vp, _vp, _vp],
    "ddh_ell_terms_create": [_hp, _i, _i, _i, _i, _i, _ip, _ip, _i, _dp, _ip],
    "ddh_ell_terms_apply": [_h, _vp, _vp, _vp],
    "ddh_dense_inverse_create": [_hp, _i, _ip, _i, _dp, _dp, C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)],
    "ddh_dense_inverse_elements": [_h, C.POINTER(_l)],
    "ddh_dense_inverse_compute": [_h, _d, _d, _vp, _ip, _vp],
    "ddh_cgemv_batch_mats": [_h, C.POINTER(_vp)],
    "ddh_ell_terms_create_dense": [_hp, _i, _i, _i, _i],
    "ddh_ell_terms_mats": [_h, C.POINTER(_vp)],
    "ddh_ell_terms_prune": [_h, _vp],
    "ddh_ell_blocks_from_dense": [_vp, _vp, _i, _i, _i, _vp],
    "ddh_ell_terms_apply_acc": [_h, _vp, _vp, _i, _vp],
    "ddh_ellband_create": [_hp, _i, _i, _i, _i, _i, _i, _i, _l, _ip, _ip, _ip, C.POINTER(_l), C.POINTER(_l), _dp, _dp, _dp, _dp],
    "ddh_ellband_factor": [_h, _i, _d, _d, _ip, _vp],
    "ddh_ellband_solve": [_h, _i, _vp, _vp, _vp],
    "ddh_ellband_info": [_h, _ip, _ip, C.POINTER(_l)],
    "ddh_ellband_gather_complex_inverse": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ddh_ellband_bordered_inverse": [_vp, _i, _i, _vp, _vp, _d, _d, _d, _d, _vp, _vp],
    "ddh_rfft_forward": [_h, _vp, _vp, _l, _l, _vp],
    "ddh_rfft_backward": [_h, _vp, _vp, _l, _l, _vp],
    "ddh_rfft_backward_deriv": [_h, _vp, _vp, _l, _l, _d, _vp],
    "ddh_rfft_backward_dual": [_h, _vp, _vp, _vp, _l, _l, _d, _vp],
    "ddh_cheb_backward_dual": [_h, _vp, _vp, _vp, _vp, _l, _l, _vp],
    "ddh_rfft_bilinear_fused": [_h, _i, C.POINTER(_vp), _dp, _i, C.POINTER(_vp), _dp, _i, C.POINTER(_vp), _l, _i,
                                _ip, _ip, _ip, _dp, _vp],
    "ddh_plan_cfft": [_hp, _i, _i],
    "ddh_cfft_forward": [_h, _vp, _vp, _l, _l, _vp],
    "ddh_cfft_backward": [_h, _vp, _vp, _l, _l, _vp],
    "ddh_plan_cheb": [_hp, _i, _i, _i, _ip, _dp],
    "ddh_cheb_forward": [_h, _vp, _vp, _l, _l, _vp],
    "ddh_cheb_forward_tiled": [_h, _vp, _vp, _l, _l, _l, _vp],
    "ddh_fft_set_stage_layout": [_h, _l],
    "ddh_fft_set_stage_block": [_h, _i],
    "ddh_fft_set_stage_window": [_h, _i, _i],
    "ddh_fft_wave_launches": [C.POINTER(_l)],
    "ddh_cheb_backward": [_h, _vp, _vp, _l, _l, _vp],
    "ddh_plan_mmt": [_hp, _i, _i, _dp],
    "ddh_mmt_apply": [_h, _vp, _vp, _l, _l, _vp],
    "ddh_lincomb": [_vp, _i, C.POINTER(_vp), _dp, _l, _vp],
    "ddh_grid_bilinear": [_vp, _i, _vp, _vp, _l, _i, _ip, _ip, _ip, _dp, _vp],
    "ddh_grid_cfl": [_vp, _vp, _i, _l, C.POINTER(_vp), _ip, C.POINTER(_l), _i, _vp],
    "ddh_grid_cfl_spherical": [_vp, _vp, _l, _i, _vp, _vp, _vp],
    "ddh_grid_reduce": [_vp, _vp, _l, _vp, _vp],
    "ddh_pencil_create": [_hp, C.POINTER(PencilGeom)],
    "ddh_pencil_add_matrix": [_h, C.POINTER(PolyMat), _i, _ip],
    "ddh_pencil_matvec": [_h, _i, _vp, _vp, _vp],
    "ddh_pencil_matvec_update": [_h, _i, _vp, _vp, _vp],
    "ddh_pencil_add_upper_bands": [_h, _i, _i, _ip, _dp, _ip],
    "ddh_pencil_matvec_solve": [_h, _i, _i, _vp, _vp, _vp],
    "ddh_pencil_factor": [_h, _i, _i, _d, _d, _ip, _ip, _i, _i, _i,
                          C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte), _i, _ip, _vp],
    "ddh_pencil_factor_real": [_h, _i, _i, _d, _d, _ip, _ip, _i, _i, _i,
                               C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte),
                               C.POINTER(C.c_ubyte), _i, _ip, _vp],
    "ddh_pencil_solve": [_h, _i, _vp, _vp, _vp],
    "ddh_pencil_set_solve_variant": [_h, _i, _i, _i],
    "ddh_pencil_set_pairing": [_h, _ip, _ip, _l],
    "ddh_pencil_set_row_blocks": [_h, _i],
    "ddh_pencil_solve_lincomb": [_h, _i, _i, C.POINTER(_vp), _dp, _vp, _vp],
    "ddh_pencil_solve_recombined": [_h, _i, _i, C.POINTER(_vp), _dp, _i, _vp, _vp, _vp],
    "ddh_pencil_solve_recombined_sparse": [_h, _i, _i, C.POINTER(_vp), _dp, _i, _vp, _vp, _vp, _vp, _vp],
    "ddh_pencil_solve_recombined_tiled": [_h, _i, _i, C.POINTER(_vp), _dp, _i, _vp, _vp, _vp, _vp, _vp],
    "ddh_pencil_matvec_update_tiled": [_h, _i, _vp, _vp, _vp],
    "ddh_pencil_flagged": [_h, _i, _ip, C.POINTER(_l), _i],
    "ddh_pencil_set_dense_inverse": [_h, _i, _dp],
    "ddh_pencil_set_dense_inverse_dev": [_h, _i, _i, _vp, _i, _vp],
    "ddh_pencil_set_block_inverse": [_h, _i, _vp],
    "ddh_pencil_lu_bytes": [_h, _i, C.POINTER(C.c_size_t)],
    "ddh_pencil_lu_info": [_h, _i, _ip],
    "ddh_pencil_lu_row_widths": [_h, _i, _ip],
    "ddh_comm_probe": [],
    "ddh_comm_unique_id": [C.POINTER(C.c_ubyte)],
    "ddh_comm_create": [_hp, _i, _i, C.POINTER(C.c_ubyte)],
    "ddh_comm_create_loopback": [_hp, _i, _i],
    "ddh_comm_info": [_h, _ip, _ip],
    "ddh_comm_allreduce": [_h, _vp, _l, _i, _vp],
    "ddh_comm_alltoall": [_h, _vp, _vp, _l, _vp],
    "ddh_comm_alltoall_part": [_h, _vp, _vp, _l, _l, _i, _l, _vp],
    "ddh_a2a_plan": [_hp, _h, _l, _l, _l, _l],
    "ddh_a2a_plan_blocks": [_hp, _h, _l, _l, _l, _l, _l, _l],
    "ddh_a2a_localize_rows": [_h, _vp, _vp, _vp],
    "ddh_a2a_localize_columns": [_h, _vp, _vp, _vp],
    "ddh_a2a_forward": [_h, _vp, _vp, _vp],
    "ddh_a2a_backward": [_h, _vp, _vp, _vp],
    "ddh_a2a_pack": [_vp, _vp, _l, _l, _l, _l, _i, _vp],
    "ddh_a2a_unpack": [_vp, _vp, _l, _l, _l, _l, _i, _vp],
    "ddh_a2av_pack": [_vp, _vp, _l, _l, _l, _i, _vp],
    "ddh_a2av_unpack": [_vp, _vp, _l, _l, _l, _i, _vp],
    "ddh_a2av_pack_b": [_vp, _vp, _l, _l, _l, _i, _l, _vp],
    "ddh_a2av_unpack_b": [_vp, _vp, _l, _l, _l, _i, _l, _vp],
}


class MmtGroup(C.Structure):
    """ddh_mmt_group of include/dedalus_hip.h"""
    _fields_ = [("mat", C.c_int), ("g_start", C.c_int), ("c_start", C.c_int), ("count", C.c_int),
                ("ell_start", C.c_int), ("ell_step", C.c_int), ("n_ell", C.c_int)]


def load(build_if_missing=False):
    """Load the shared library (no device needed). Raises if it cannot be found."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch ships its own libamdhip64: it must be the HIP runtime this library binds to, so that
        # device pointers and streams are shared (loading ours first would create a second runtime)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from . import build
            build.build_library()
        else:
            raise DdhError("libdedalus_hip.so not found at %s -- run `python -m dedalus_amd.build` "
                           "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        if os.environ.get("DDH_LIB") and not hasattr(lib, name):
            continue                     # an older experimental build: calling a missing entry point raises
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.ddh_last_error.argtypes = []
    lib.ddh_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().ddh_last_error().decode("utf-8", "replace")
        raise DdhError("%s failed (status %d): %s" % (what or "libdedalus_hip call", status, msg))


# Optional measurement hooks (tools/bench_configs.py): `profiler(name, fn, args) -> status` wraps every entry point
# (HIP events per call), `cost_log(name, flops, bytes)` receives the ALGORITHMIC work a wrapper hands to an entry point
# (dedalus_amd/executor.py::note_cost).  Both are None in ordinary runs: one attribute test per call.
profiler = None
cost_log = None


def call(name, *args):
    lib = load()
    if profiler is not None:
        check(profiler(name, getattr(lib, name), args), name)
        return
    check(getattr(lib, name)(*args), name)


def note_cost(name, flops, nbytes):
    if cost_log is not None:
        cost_log(name, float(flops), float(nbytes))


def as_dp(a):
    return a.ctypes.data_as(_dp)


def as_ip(a):
    return a.ctypes.data_as(_ip)


def as_bp(a):
    return a.ctypes.data_as(C.POINTER(C.c_byte))


def as_ubp(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))
