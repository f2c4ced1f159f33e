"""Host-side mirror of the reference's viewport type for the hot path.

FractalState / FractalType / Presets follow src/fractal_state.h (same field names,
same defaults), restricted to the fields the Mandelbrot and Julia compute path reads
(the union packed by ComputeEffect::update_from_state, src/compute_effect_manager.h:84-140).
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, fields

import numpy as np

from . import _capi


class FractalType(enum.IntEnum):
    """src/fractal_state.h:6-14.  Mandelbrot and JuliaSet are the hot path; BurningShip and Deep_Zoom are
    its variants (section 8 f1/f4); Mandelbulb and Phoenix are rejected with FR_ERR_UNSUPPORTED."""
    Mandelbrot = 0
    JuliaSet = 1
    BurningShip = 2
    Mandelbulb = 3
    Phoenix = 4
    Deep_Zoom = 5


class Precision(enum.IntEnum):
    F32 = _capi.FR_PRECISION_F32   # what the reference's shaders compute in
    F64 = _capi.FR_PRECISION_F64


_F32 = lambda v: float(np.float32(v))  # noqa: E731  (the reference stores these as float)


@dataclass
class FractalState:
    """src/fractal_state.h:16-91 (hot-path fields, reference defaults)."""
    center_x: float = -0.5                      # :18
    center_y: float = 0.0                       # :19
    zoom: float = 3.0                           # :20
    max_iterations: int = 256                   # :21
    julia_c_real: float = _F32(-0.7)            # :29
    julia_c_imag: float = _F32(0.27015)         # :30
    bailout: float = 4.0                        # :36
    antialiasing_samples: int = 1               # :37
    palette_mode: int = 0                       # :40
    color_offset: float = 0.0                   # :41
    color_scale: float = 1.0                    # :42
    interior_style: int = 0                     # :47
    orbit_trap_enabled: bool = False            # :48
    orbit_trap_radius: float = 0.5              # :49
    stripe_enabled: bool = False                # :50
    stripe_density: float = 10.0                # :51
    color_brightness: float = 1.0               # :77
    color_saturation: float = 1.0               # :78
    color_contrast: float = 1.0                 # :79
    use_perturbation: bool = False              # :86  (Deep_Zoom: compute and use the fp64 reference orbit)

    def reset(self) -> None:
        """FractalState::reset(), src/fractal_state.h:135-153 (note zoom 1.5, not 3.0)."""
        self.center_x, self.center_y, self.zoom, self.max_iterations = -0.5, 0.0, 1.5, 256
        self.color_brightness = self.color_saturation = self.color_contrast = 1.0

    # -- C ABI conversion ---------------------------------------------------------------
    def to_params(self, fractal_type: FractalType = FractalType.Mandelbrot,
                  precision: Precision = Precision.F64, post_chain: bool = False) -> _capi.fr_params:
        p = _capi.fr_params()
        p.fractal_type = int(fractal_type)
        p.precision = int(precision)
        for f in fields(self):
            v = getattr(self, f.name)
            setattr(p, f.name, int(v) if isinstance(v, bool) else v)
        p.flags = _capi.FR_FLAG_POST_CHAIN if post_chain else 0
        return p

    @classmethod
    def from_params(cls, p: _capi.fr_params) -> "FractalState":
        kw = {}
        for f in fields(cls):
            v = getattr(p, f.name)
            kw[f.name] = bool(v) if f.type == "bool" else v
        return cls(**kw)


@dataclass(frozen=True)
class Preset:
    name: str
    type: FractalType
    center_x: float
    center_y: float
    zoom: float
    iterations: int


# Presets::MANDELBROT_PRESETS, src/fractal_state.h:171-180
MANDELBROT_PRESETS = (
    Preset("Overview", FractalType.Mandelbrot, -0.5, 0.0, 2.5, 256),
    Preset("Seahorse Valley", FractalType.Mandelbrot, -0.743643887037151, 0.13182590420533, 0.008, 1024),
    Preset("Elephant Valley", FractalType.Mandelbrot, 0.257, 0.0, 0.015, 768),
    Preset("Triple Spiral", FractalType.Mandelbrot, -0.088, 0.654, 0.02, 512),
    Preset("Mini Mandelbrot", FractalType.Mandelbrot, -1.7497, 0.00001, 0.0005, 1024),
    Preset("Spiral Galaxy", FractalType.Mandelbrot, -0.7453, 0.1127, 0.01, 768),
)

# DeepZoomPresets::createSeahorseZoom, src/deep_zoom_system.cpp:576-583 (the C4 benchmark view)
SEAHORSE_DEEP = Preset("Seahorse deep", FractalType.Mandelbrot, -0.743643887037151, 0.13182590420533, 1e-6, 16384)


def pack_push_constants(state: FractalState, fractal_type: FractalType) -> np.ndarray:
    """ComputeEffect::update_from_state (src/compute_effect_manager.h:84-140): the 80-byte
    ComputePushConstants block as 20 float32."""
    out = (_capi.C.c_float * 20)()
    p = state.to_params(fractal_type)
    _capi.check(_capi.lib().fr_pack_push_constants(_capi.C.byref(p), out))
    return np.array(out[:], dtype=np.float32)
