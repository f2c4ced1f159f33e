"""ctypes front end of the CPU oracle (oracle/fr_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from the product package.
PARITY PIN: no tests or golden vectors exist upstream and no Vulkan driver runs in
this image; the fp32 variants are pinned against vectors obtained by executing the
reference's compiled shaders (tests/golden/spv_frames.npz) -- see fr_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, fields

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfr_oracle.so")


class _FroParams(C.Structure):
    _fields_ = [
        ("fractal", C.c_int32), ("precision", C.c_int32),
        ("center_x", C.c_double), ("center_y", C.c_double), ("zoom", C.c_double),
        ("max_iterations", C.c_int32), ("bailout", C.c_float),
        ("julia_c_real", C.c_double), ("julia_c_imag", C.c_double),
        ("aa", C.c_int32), ("palette_mode", C.c_int32),
        ("color_offset", C.c_float), ("color_scale", C.c_float),
        ("interior_style", C.c_int32), ("orbit_trap_enabled", C.c_int32),
        ("orbit_trap_radius", C.c_float), ("stripe_enabled", C.c_int32),
        ("stripe_density", C.c_float),
        ("brightness", C.c_float), ("saturation", C.c_float), ("contrast", C.c_float),
        ("post_chain", C.c_int32), ("use_perturbation", C.c_int32),
    ]


@dataclass
class OracleParams:
    """Defaults are the FractalState initialisers (src/fractal_state.h:18-51,77-79)."""
    fractal: int = 0
    precision: int = 1
    center_x: float = -0.5
    center_y: float = 0.0
    zoom: float = 3.0
    max_iterations: int = 256
    bailout: float = 4.0
    julia_c_real: float = float(np.float32(-0.7))
    julia_c_imag: float = float(np.float32(0.27015))
    aa: int = 1
    palette_mode: int = 0
    color_offset: float = 0.0
    color_scale: float = 1.0
    interior_style: int = 0
    orbit_trap_enabled: int = 0
    orbit_trap_radius: float = 0.5
    stripe_enabled: int = 0
    stripe_density: float = 10.0
    brightness: float = 1.0
    saturation: float = 1.0
    contrast: float = 1.0
    post_chain: int = 0
    use_perturbation: int = 1

    def to_c(self) -> _FroParams:
        c = _FroParams()
        for f in fields(self):
            setattr(c, f.name, getattr(self, f.name))
        return c


def build(force: bool = False) -> str:
    """Compile the oracle with gcc via oracle/Makefile."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("fr_oracle.c", "fr_oracle_sample.inc", "fr_oracle.h"))
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.fro_render_rows.restype = C.c_int64
        L.fro_render_rows.argtypes = [C.POINTER(_FroParams), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.fro_palette.argtypes = [C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_float)]
        L.fro_post_chain.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_int32]
        L.fro_pack_push_constants.argtypes = [C.POINTER(_FroParams), C.POINTER(C.c_float)]
        L.fro_reference_orbit.restype = C.c_int32
        L.fro_reference_orbit.argtypes = [C.c_double, C.c_double, C.c_int32, C.c_void_p]
        L.fro_export_rgb8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        L.fro_export8_scan.restype = C.c_int64
        L.fro_export8_scan.argtypes = [C.c_void_p]
        L.fro_colorize.argtypes = [C.POINTER(_FroParams), C.c_int64, C.c_void_p, C.c_void_p]
        L.fro_colorize.restype = None
        L.fro_max_threads.restype = C.c_int32
        _lib = L
    return _lib


@dataclass
class OracleFrame:
    rgba: np.ndarray      # (rows, W, 4) float32
    nu: np.ndarray        # (rows, W) float64
    iter: np.ndarray      # (rows, W) int32
    zre: np.ndarray
    zim: np.ndarray
    executed: int         # exact number of executed iterations


def render(p: OracleParams, W: int, H: int, y0: int = 0, y1: int | None = None,
           threads: int = 0, planes: bool = True) -> OracleFrame:
    y1 = H if y1 is None else y1
    rows = y1 - y0
    rgba = np.empty((rows, W, 4), np.float32)
    if planes:
        nu = np.empty((rows, W), np.float64)
        it = np.empty((rows, W), np.int32)
        zre = np.empty((rows, W), np.float64)
        zim = np.empty((rows, W), np.float64)
        ptrs = [a.ctypes.data for a in (rgba, nu, it, zre, zim)]
    else:
        nu = it = zre = zim = None
        ptrs = [rgba.ctypes.data, None, None, None, None]
    cp = p.to_c()
    n = lib().fro_render_rows(C.byref(cp), W, H, y0, y1, *ptrs, threads)
    return OracleFrame(rgba, nu, it, zre, zim, int(n))


def palette(shader: int, mode: int, t: float) -> np.ndarray:
    out = (C.c_float * 3)()
    lib().fro_palette(shader, mode, t, out)
    return np.array(out[:], np.float32)


def post_chain(rgb, brightness=1.0, saturation=1.0, contrast=1.0, julia_floors=0) -> np.ndarray:
    buf = (C.c_float * 3)(*[float(v) for v in rgb])
    lib().fro_post_chain(buf, brightness, saturation, contrast, julia_floors)
    return np.array(buf[:], np.float32)


def pack_push_constants(p: OracleParams) -> np.ndarray:
    out = (C.c_float * 20)()
    cp = p.to_c()
    lib().fro_pack_push_constants(C.byref(cp), out)
    return np.array(out[:], np.float32)


def reference_orbit(cx: float, cy: float, max_iter: int) -> np.ndarray:
    buf = np.zeros((max_iter, 2), np.float64)
    n = lib().fro_reference_orbit(cx, cy, max_iter, buf.ctypes.data)
    return buf[:n].copy()


def export_rgb8(rgba: np.ndarray, through_half: bool = False) -> np.ndarray:
    H, W = rgba.shape[:2]
    src = np.ascontiguousarray(rgba, np.float32)
    out = np.empty((H, W, 3), np.uint8)
    lib().fro_export_rgb8(src.ctypes.data, W, H, out.ctypes.data, int(through_half))
    return out


def export8_scan():
    """(violations, first): the 8-bit export's byte scanned over every float of [0, 1] (fro_export8_scan)."""
    first = np.empty(256, np.uint32)
    bad = int(lib().fro_export8_scan(first.ctypes.data))
    return bad, first


def colorize(p: "OracleParams", nu: np.ndarray) -> np.ndarray:
    """Colour stage alone (fro_colorize): nu plane -> (..., 4) float32 RGBA."""
    src = np.ascontiguousarray(nu, np.float64)
    out = np.empty(src.shape + (4,), np.float32)
    cp = p.to_c()
    lib().fro_colorize(C.byref(cp), src.size, src.ctypes.data, out.ctypes.data)
    return out


def max_threads() -> int:
    return int(lib().fro_max_threads())
