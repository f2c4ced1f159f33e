"""ctypes binding of libfractalrenderer_amd.so (include/fractalrenderer_amd.h).

The shared library is the product; this module only declares its C ABI to Python.
There is no fallback of any kind: if the library is missing, import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FR_LIB_PATH: load another build of the same library (A/B runs of two kernel versions in one GPU session)
LIB_PATH = os.environ.get("FR_LIB_PATH") or os.path.join(_HERE, "libfractalrenderer_amd.so")

FR_OK = 0
FR_ERR_INVALID_ARG = -1
FR_ERR_NO_DEVICE = -2
FR_ERR_HIP = -3
FR_ERR_UNSUPPORTED = -4
FR_ERR_IO = -5
FR_ERR_PARSE = -6
FR_ERR_NOMEM = -7
FR_ERR_INTERNAL = -8

FR_MEM_DEVICE = 0
FR_MEM_HOST = 1

FR_LAYOUT_PACKED = 0
FR_LAYOUT_FRAME = 1

FR_GATHER_AUTO = 0
FR_GATHER_PEER = 1
FR_GATHER_RCCL = 2
FR_ROOT_ROTATE = -1

FR_PRECISION_F32 = 0
FR_PRECISION_F64 = 1

FR_FLAG_POST_CHAIN = 0x1


class fr_params(C.Structure):
    _fields_ = [
        ("fractal_type", C.c_int32), ("precision", C.c_int32),
        ("center_x", C.c_double), ("center_y", C.c_double), ("zoom", C.c_double),
        ("max_iterations", C.c_int32), ("bailout", C.c_float),
        ("julia_c_real", C.c_double), ("julia_c_imag", C.c_double),
        ("antialiasing_samples", C.c_int32), ("palette_mode", C.c_int32),
        ("color_offset", C.c_float), ("color_scale", C.c_float),
        ("interior_style", C.c_int32),
        ("orbit_trap_enabled", C.c_int32), ("orbit_trap_radius", C.c_float),
        ("stripe_enabled", C.c_int32), ("stripe_density", C.c_float),
        ("color_brightness", C.c_float), ("color_saturation", C.c_float), ("color_contrast", C.c_float),
        ("flags", C.c_uint32), ("use_perturbation", C.c_int32),
    ]


class fr_output(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("nu", C.c_void_p), ("iter", C.c_void_p), ("memory", C.c_int32), ("layout", C.c_int32)]


class fr_shard(C.Structure):
    _fields_ = [("part", C.c_uint32), ("nparts", C.c_uint32), ("rows_per_strip", C.c_uint32)]


class fr_png_text(C.Structure):
    _fields_ = [("key", C.c_char_p), ("text", C.c_char_p)]


class fr_zoom_keyframe(C.Structure):
    _fields_ = [("center_x", C.c_double), ("center_y", C.c_double), ("zoom", C.c_double), ("duration", C.c_float)]


class fr_anim_info(C.Structure):
    _fields_ = [("duration", C.c_float), ("loop", C.c_int32), ("target_fps", C.c_int32),
                ("export_width", C.c_int32), ("export_height", C.c_int32), ("keyframe_count", C.c_int32)]


FR_FRAME_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_int32, C.c_void_p)


class fr_anim_render_options(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("first_frame", C.c_int32), ("frame_count", C.c_int32),
                ("frame_step", C.c_int32), ("max_iterations_override", C.c_int32), ("fractal_type_override", C.c_int32),
                ("on_frame_complete", FR_FRAME_CALLBACK), ("user", C.c_void_p), ("raw_fd", C.c_int32), ("reserved", C.c_int32)]


class fr_keyframe(C.Structure):
    _fields_ = [("time", C.c_float), ("interp_type", C.c_int32), ("state", fr_params)]


# every symbol include/fractalrenderer_amd.h declares: name -> (restype, argtypes)
_P = C.POINTER
SIGNATURES = {
    "fr_params_default": (C.c_int, [_P(fr_params)]),
    "fr_params_reset": (C.c_int, [_P(fr_params)]),
    "fr_params_validate": (C.c_int, [_P(fr_params), C.c_uint32, C.c_uint32]),
    "fr_pack_push_constants": (C.c_int, [_P(fr_params), _P(C.c_float)]),
    "fr_ctx_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "fr_ctx_destroy": (None, [C.c_void_p]),
    "fr_shard_rows": (C.c_uint32, [_P(fr_shard), C.c_uint32]),
    "fr_shard_global_row": (C.c_uint32, [_P(fr_shard), C.c_uint32, C.c_uint32]),
    "fr_render": (C.c_int, [C.c_void_p, _P(fr_params), C.c_uint32, C.c_uint32, _P(fr_output)]),
    "fr_render_shard": (C.c_int, [C.c_void_p, _P(fr_params), C.c_uint32, C.c_uint32, _P(fr_shard), _P(fr_output)]),
    "fr_render_shard_async": (C.c_int, [C.c_void_p, _P(fr_params), C.c_uint32, C.c_uint32, _P(fr_shard),
                                        _P(fr_output), C.c_void_p]),
    "fr_ctx_reserve": (C.c_int, [C.c_void_p, _P(fr_params), C.c_uint32, C.c_uint32, _P(fr_shard)]),
    "fr_ctx_check": (C.c_int, [C.c_void_p]),
    "fr_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "fr_node_create": (C.c_int, [_P(C.c_int), C.c_int, _P(C.c_void_p)]),
    "fr_node_destroy": (None, [C.c_void_p]),
    "fr_node_device_count": (C.c_int, [C.c_void_p]),
    "fr_node_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "fr_node_render": (C.c_int, [C.c_void_p, _P(fr_params), C.c_uint32, C.c_uint32, C.c_int, _P(fr_output)]),
    "fr_node_render_async": (C.c_int, [C.c_void_p, _P(fr_params), C.c_uint32, C.c_uint32, C.c_int, _P(fr_output)]),
    "fr_node_wait": (C.c_int, [C.c_void_p]),
    "fr_node_submit": (C.c_int, [C.c_void_p, _P(fr_params), C.c_uint32, C.c_uint32, C.c_int, _P(fr_output), _P(C.c_uint64)]),
    "fr_node_wait_frame": (C.c_int, [C.c_void_p, C.c_uint64]),
    "fr_node_in_flight": (C.c_int, [C.c_void_p]),
    "fr_node_render_animation": (C.c_int, [C.c_void_p, C.c_void_p, _P(fr_params), C.c_void_p, C.c_char_p, _P(C.c_int32)]),
    "fr_node_last_gather": (C.c_int, [C.c_void_p]),
    "fr_node_last_kernel_ms": (C.c_float, [C.c_void_p, C.c_int]),
    "fr_ctx_last_kernel_ms": (C.c_float, [C.c_void_p]),
    "fr_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "fr_ctx_last_grid": (C.c_int, [C.c_void_p]),
    "fr_ctx_compute_units": (C.c_int, [C.c_void_p]),
    "fr_colorize_supported": (C.c_int, [_P(fr_params)]),
    "fr_colorize_async": (C.c_int, [C.c_void_p, _P(fr_params), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fr_export_rgb8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32, C.c_int32]),
    "fr_export_rgb16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32, C.c_int32]),
    "fr_export_rgb8_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p]),
    "fr_export_rgb16_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p]),
    "fr_write_png": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, _P(fr_png_text), C.c_int32, C.c_int32]),
    "fr_write_raw_rgb24": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32]),
    "fr_frame_path": (C.c_int, [C.c_char_p, C.c_int32, C.c_char_p, C.c_size_t]),
    "fr_render_frame_png": (C.c_int, [C.c_void_p, _P(fr_params), C.c_uint32, C.c_uint32, C.c_char_p]),
    "fr_anim_load": (C.c_int, [C.c_char_p, _P(C.c_void_p)]),
    "fr_anim_parse": (C.c_int, [C.c_char_p, C.c_size_t, _P(C.c_void_p)]),
    "fr_anim_free": (None, [C.c_void_p]),
    "fr_anim_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "fr_anim_get_info": (C.c_int, [C.c_void_p, _P(fr_anim_info)]),
    "fr_anim_get_keyframe": (C.c_int, [C.c_void_p, C.c_int32, _P(fr_keyframe)]),
    "fr_anim_name": (C.c_char_p, [C.c_void_p]),
    "fr_anim_description": (C.c_char_p, [C.c_void_p]),
    "fr_anim_create": (C.c_int, [_P(C.c_void_p)]),
    "fr_anim_add_keyframe": (C.c_int, [C.c_void_p, C.c_float, _P(fr_params), C.c_int32]),
    "fr_anim_state_at": (C.c_int, [C.c_void_p, C.c_float, _P(fr_params), _P(fr_params)]),
    "fr_anim_frame_count": (C.c_int32, [C.c_void_p]),
    "fr_anim_frame_time": (C.c_float, [C.c_void_p, C.c_int32]),
    "fr_zoom_path_create": (C.c_int, [_P(C.c_void_p)]),
    "fr_zoom_path_free": (None, [C.c_void_p]),
    "fr_zoom_path_play": (C.c_int, [C.c_void_p, _P(fr_zoom_keyframe), C.c_int32]),
    "fr_zoom_path_zoom_to": (C.c_int, [C.c_void_p, _P(fr_params), C.c_double, C.c_double, C.c_double, C.c_float]),
    "fr_zoom_path_update": (C.c_int, [C.c_void_p, C.c_float, _P(fr_params), _P(C.c_int32), _P(C.c_float), _P(C.c_int32)]),
    "fr_zoom_preset": (C.c_int, [C.c_int32, _P(fr_zoom_keyframe)]),
    "fr_reference_orbit": (C.c_int, [C.c_double, C.c_double, C.c_int32, C.c_void_p, _P(C.c_int32)]),
    "fr_last_error": (C.c_char_p, []),
    "fr_status_string": (C.c_char_p, [C.c_int]),
    "fr_version": (None, [_P(C.c_int), _P(C.c_int)]),
}


# internal (fractalrenderer_amd/csrc/fr_tuning.h): queue / stream tuning for tests, tools and A/B measurements
INTERNAL_SIGNATURES = {
    "fr_ctx_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "fr_export8_thresholds": (None, [_P(C.c_float)]),          # fr_internal.h: byte thresholds of the 8-bit export
    "fr_export8_thresholds_host_powf": (None, [_P(C.c_float)]),
    "fr_node_rccl_selftest": (C.c_int, [C.c_int, C.c_size_t, _P(C.c_int)]),
    "fr_ctx_last_pool_closing": (C.c_int, [C.c_void_p]),
    "fr_node_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "fr_node_rccl_usable": (C.c_int, [C.c_void_p]),
    "fr_node_mapped_runtimes": (C.c_int, [C.c_char_p, C.c_size_t]),
}
PUBLIC_OPTIONS = ("periodicity", "staging", "shards", "tile_kernel", "timing", "diag_buffer", "diag_stride")
TUNING_NAMES = ("workgroups_per_cu", "run_max", "run_min", "shift_bias", "stage_first", "pool_refill_at", "stream_run_max",
                "stream_run_min", "stream_workgroups_per_cu", "probes", "stream_probes", "regions", "stream_rotate",
                "tile_pixels", "tile_exit", "tile_exit_from", "prepare", "pool_items_per_wg", "subtile_shape", "debug_region_blocks", "debug_prologue_epoch", "ssaa", "ssaa_band_samples", "stripes")


class FractalRendererError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"fractalrenderer_amd: status {status}: {message}")
        self.status = status


_lib = None


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64 (SONAME
    libamdhip64.so.7, the same as /opt/rocm's, but requested by torch under the name libamdhip64.so).  If
    this library is loaded BEFORE torch it binds /opt/rocm's copy, torch then loads its own next to it, and
    the second runtime to initialise finds no device ("No HIP GPUs are available").  Loading torch's copy
    first -- by path, without importing torch -- makes both bind the same runtime whatever the import
    order.  Without an installed torch this is a no-op and the library uses /opt/rocm's runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for loc in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return


def lib() -> C.CDLL:
    """Load the C-ABI library.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C fractalrenderer_amd/csrc`. There is no Python/CPU fallback for the render path.")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in INTERNAL_SIGNATURES.items():
            # internal entry points: required of the in-tree build, optional for an older build loaded through FR_LIB_PATH
            # (A/B runs of two kernel versions in one GPU session)
            if not hasattr(L, name) and os.environ.get("FR_LIB_PATH"):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int) -> int:
    if status < 0:
        raise FractalRendererError(status, lib().fr_last_error().decode("utf-8", "replace"))
    return status
