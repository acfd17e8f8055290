"""ctypes binding of libnemo_hip.so (C ABI declared in include/nemo_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call
fails, a NemoHipError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NEMO_HIP_LIB selects a diagnostic build of the same library (e.g. the phase-stamp build).
LIB_PATH = os.environ.get('NEMO_HIP_LIB') or os.path.join(_HERE, 'lib', 'libnemo_hip.so')

EXPORTS = ['nemo_ctx_create', 'nemo_ctx_destroy', 'nemo_last_error', 'nemo_mesh_register', 'nemo_workspace_reserve', 'nemo_solve_static',
           'nemo_set_schedule', 'nemo_campos_to_R_T', 'nemo_rasterize', 'nemo_interpolate',
           'nemo_interpolate_backward', 'nemo_clutter_score', 'nemo_score_fwd_bwd',
           'nemo_grid_search', 'nemo_grid_prerender', 'nemo_solve', 'nemo_solve6', 'nemo_feature_head', 'nemo_bytes_per_covered_pixel', 'nemo_counter_reset',
           'nemo_counter_read', 'nemo_grid_store_kind', 'nemo_solve_table_capacity',
           'nemo_solve_last_timing', 'nemo_solve_table_stats']


class NemoHipError(RuntimeError):
    pass


class NemoConfig(C.Structure):
    _fields_ = [('raster_size', C.c_int32), ('win_h', C.c_int32), ('win_w', C.c_int32),
                ('win_r0', C.c_int32), ('win_c0', C.c_int32), ('feat_dim', C.c_int32),
                ('ndc_scale', C.c_float), ('perspective_correct', C.c_int32),
                ('clutter', C.c_int32), ('k_eps', C.c_double)]


_lib = None


def load():
    """Load libnemo_hip.so; raises NemoHipError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NemoHipError(
            f'{LIB_PATH} not found: build it with nemo_amd/csrc/build.sh (hipcc, gfx950). '
            'There is no CPU fallback for the pose-solve path.')
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.nemo_last_error.restype = C.c_char_p
    lib.nemo_last_error.argtypes = [vp]
    lib.nemo_ctx_create.argtypes = [C.POINTER(NemoConfig), C.POINTER(vp)]
    lib.nemo_ctx_destroy.argtypes = [vp]
    lib.nemo_mesh_register.argtypes = [vp, vp, i32, vp, i32, vp]
    lib.nemo_set_schedule.argtypes = [vp, vp, i32, f32, f32, f32]
    lib.nemo_campos_to_R_T.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
    lib.nemo_rasterize.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.nemo_interpolate.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    lib.nemo_interpolate_backward.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp]
    lib.nemo_clutter_score.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    lib.nemo_score_fwd_bwd.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.nemo_grid_search.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.nemo_grid_prerender.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, vp]
    lib.nemo_solve.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.nemo_solve_static.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.nemo_workspace_reserve.argtypes = [vp, i32, i32]
    lib.nemo_feature_head.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, vp, vp]
    lib.nemo_solve6.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.nemo_bytes_per_covered_pixel.restype = C.c_int64
    lib.nemo_bytes_per_covered_pixel.argtypes = [vp]
    lib.nemo_counter_reset.argtypes = [vp, vp]
    lib.nemo_counter_read.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.nemo_grid_store_kind.argtypes = [vp]
    lib.nemo_solve_table_capacity.restype = C.c_int64
    lib.nemo_solve_table_capacity.argtypes = [vp]
    lib.nemo_solve_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.nemo_solve_table_stats.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int:
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(ctx, rc, what):
    if rc < 0:
        msg = load().nemo_last_error(ctx)
        raise NemoHipError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')
    return rc
