"""fractalrenderer_amd -- MI355X-native escape-time renderer (Mandelbrot/Julia hot path).

The product is libfractalrenderer_amd.so (C ABI: include/fractalrenderer_amd.h; HIP kernels:
fractalrenderer_amd/csrc/).  This package is the thin host-side mirror of the reference's
interface for that path.  Importing it loads the library and fails loudly if it is absent.
"""
from . import _capi
from ._capi import FractalRendererError, lib
from .state import (FractalState, FractalType, Precision, Preset, MANDELBROT_PRESETS,
                    SEAHORSE_DEEP, pack_push_constants)
from .renderer import (Renderer, Node, Shard, write_png, write_raw_rgb24, frame_path, export8_thresholds, rccl_selftest,
                       mapped_runtimes)
from .animation import (AnimationSystem, AnimationRenderer, InterpolationType, Keyframe, DeepZoomPath, ZoomKeyframe)

lib()  # no silent fallback: a missing/incomplete library is an import error

__all__ = [
    "FractalRendererError", "lib", "FractalState", "FractalType", "Precision", "Preset",
    "MANDELBROT_PRESETS", "SEAHORSE_DEEP", "pack_push_constants", "Renderer", "Node", "Shard",
    "write_png", "write_raw_rgb24", "frame_path", "export8_thresholds", "rccl_selftest", "mapped_runtimes",
    "AnimationSystem", "AnimationRenderer", "InterpolationType", "Keyframe", "DeepZoomPath", "ZoomKeyframe",
]
