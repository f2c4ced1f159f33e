"""Mirror of the reference's animation model for the hot path's callers.

  AnimationSystem   src/animation_system.{h,cpp}  (keyframes, interpolate, .franim I/O)
  AnimationRenderer src/animation_renderer.{h,cpp} (the frame loop that invokes the
                    RenderFrameCallback plug-in point once per frame)
All arithmetic (JSON parsing, easing, interpolation, frame timing) lives in the C library
(fractalrenderer_amd/csrc/fr_franim.c); these classes only hold handles.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass
from typing import Callable, List, Optional

from . import _capi
from .renderer import frame_path
from .state import FractalState, FractalType, Precision


class InterpolationType(enum.IntEnum):
    """src/animation_system.h:8-14"""
    Linear = 0
    EaseInOut = 1
    EaseIn = 2
    EaseOut = 3
    Exponential = 4


@dataclass
class Keyframe:
    """src/animation_system.h:16-22"""
    time: float
    state: FractalState
    interp_type: InterpolationType = InterpolationType.EaseInOut


class AnimationSystem:
    """src/animation_system.h:37-83.  `fractal_state` plays the role of the reference's
    FractalState& member: it is what interpolate() returns when there are no keyframes."""

    def __init__(self, fractal_state: Optional[FractalState] = None):
        self._lib = _capi.lib()
        self.fractal_state = fractal_state if fractal_state is not None else FractalState()
        h = C.c_void_p()
        _capi.check(self._lib.fr_anim_create(C.byref(h)))
        self._h = h

    def _replace(self, h: C.c_void_p) -> None:
        if self._h:
            self._lib.fr_anim_free(self._h)
        self._h = h

    def __del__(self):
        try:
            if self._h:
                self._lib.fr_anim_free(self._h)
                self._h = None
        except Exception:
            pass

    # -- Save/Load (src/animation_system.cpp:221-313) -----------------------------------
    def load_from_file(self, filename: str) -> bool:
        h = C.c_void_p()
        st = self._lib.fr_anim_load(os.fsencode(filename), C.byref(h))
        if st != _capi.FR_OK:
            return False                      # the reference returns false and logs
        self._replace(h)
        return True

    def loads(self, text: str) -> None:
        raw = text.encode("utf-8")
        h = C.c_void_p()
        _capi.check(self._lib.fr_anim_parse(raw, len(raw), C.byref(h)))
        self._replace(h)

    def save_to_file(self, filename: str) -> bool:
        return self._lib.fr_anim_save(self._h, os.fsencode(filename)) == _capi.FR_OK

    # -- keyframes ----------------------------------------------------------------------
    def add_keyframe(self, time: float, state: FractalState,
                     interp_type: InterpolationType = InterpolationType.EaseInOut) -> None:
        p = state.to_params()
        _capi.check(self._lib.fr_anim_add_keyframe(self._h, time, C.byref(p), int(interp_type)))

    def get_keyframes(self) -> List[Keyframe]:
        out = []
        for i in range(self.info.keyframe_count):
            k = _capi.fr_keyframe()
            _capi.check(self._lib.fr_anim_get_keyframe(self._h, i, C.byref(k)))
            out.append(Keyframe(k.time, FractalState.from_params(k.state), InterpolationType(k.interp_type)))
        return out

    @property
    def info(self) -> _capi.fr_anim_info:
        i = _capi.fr_anim_info()
        _capi.check(self._lib.fr_anim_get_info(self._h, C.byref(i)))
        return i

    def get_duration(self) -> float:
        return float(self.info.duration)

    @property
    def name(self) -> str:
        return self._lib.fr_anim_name(self._h).decode("utf-8")

    @property
    def description(self) -> str:
        return self._lib.fr_anim_description(self._h).decode("utf-8")

    # -- interpolation (src/animation_system.cpp:82-181) ---------------------------------
    def interpolate(self, time: float) -> FractalState:
        base = self.fractal_state.to_params()
        out = _capi.fr_params()
        _capi.check(self._lib.fr_anim_state_at(self._h, time, C.byref(base), C.byref(out)))
        return FractalState.from_params(out)

    # -- frame arithmetic (src/animation_renderer.cpp:48,80) -------------------------------
    def frame_count(self) -> int:
        return int(self._lib.fr_anim_frame_count(self._h))

    def frame_time(self, frame: int) -> float:
        return float(self._lib.fr_anim_frame_time(self._h, frame))


# bool(const FractalState&, uint32_t width, uint32_t height, const std::string& path)
RenderFrameCallback = Callable[[FractalState, int, int, str], bool]


class AnimationRenderer:
    """The frame loop of AnimationRenderer::start_render (src/animation_renderer.cpp:26-152):
    for frame in range(int(duration*fps)): state = interpolate(frame/float(fps));
    render_frame_callback(state, export_width, export_height, "<folder>/frame_%06d.png")."""

    def __init__(self, render_frame_callback: Optional[RenderFrameCallback] = None):
        self.render_frame_callback = render_frame_callback
        self.current_frame = 0
        self.total_frames = 0
        self.on_frame_complete: Optional[Callable[[int, int], None]] = None

    def start_render(self, anim_system: AnimationSystem, output_folder: str = "animation_frames",
                     frames: Optional[range] = None, width: Optional[int] = None,
                     height: Optional[int] = None) -> bool:
        if self.render_frame_callback is None:
            return False                                           # :207-210 "No render callback set"
        info = anim_system.info
        if info.keyframe_count < 2:
            return False                                           # :35-42 "Need at least 2 keyframes"
        self.total_frames = anim_system.frame_count()              # :48
        w = width or info.export_width
        h = height or info.export_height
        for frame in (frames if frames is not None else range(self.total_frames)):
            if frame >= self.total_frames:
                break
            self.current_frame = frame
            time = anim_system.frame_time(frame)                   # :80
            state = anim_system.interpolate(time)                  # :83
            path = frame_path(output_folder, frame)                # :86-88
            if not self.render_frame_callback(state, w, h, path):  # :100-107 -> :216
                return False
            if self.on_frame_complete:
                self.on_frame_complete(frame, self.total_frames)   # :124-126
        return True


@dataclass
class ZoomKeyframe:
    """src/deep_zoom_system.h:84-89 (centre / zoom are the reference's double-backed ArbitraryFloat)"""
    center_x: float = 0.0
    center_y: float = 0.0
    zoom: float = 1.0
    duration: float = 0.0


class DeepZoomPath:
    """The zoom-path animation of DeepZoomManager (src/deep_zoom_system.cpp:454-556): playZoomPath, zoomTo,
    update_animation.  `state` is the FractalState it moves (the reference's DeepZoomState centre / zoom).
    All arithmetic is in the C library (fractalrenderer_amd/csrc/fr_zoompath.c)."""

    PRESETS = ("Seahorse", "Elephant", "MiniMandelbrot")        # DeepZoomPresets, :575-601

    def __init__(self, state: Optional[FractalState] = None):
        self._lib = _capi.lib()
        self.state = state if state is not None else FractalState()
        self.zoom_animating = False
        self.zoom_progress = 0.0
        h = C.c_void_p()
        _capi.check(self._lib.fr_zoom_path_create(C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                self._lib.fr_zoom_path_free(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def preset(which) -> ZoomKeyframe:
        idx = DeepZoomPath.PRESETS.index(which) if isinstance(which, str) else int(which)
        k = _capi.fr_zoom_keyframe()
        _capi.check(_capi.lib().fr_zoom_preset(idx, C.byref(k)))
        return ZoomKeyframe(k.center_x, k.center_y, k.zoom, k.duration)

    def play_zoom_path(self, path: List[ZoomKeyframe]) -> None:
        arr = (_capi.fr_zoom_keyframe * max(1, len(path)))()
        for i, k in enumerate(path):
            arr[i].center_x, arr[i].center_y, arr[i].zoom, arr[i].duration = k.center_x, k.center_y, k.zoom, k.duration
        _capi.check(self._lib.fr_zoom_path_play(self._h, arr, len(path)))
        self.zoom_animating, self.zoom_progress = bool(path), 0.0

    def zoom_to(self, target_x: float, target_y: float, target_zoom: float, duration: float) -> None:
        p = self.state.to_params()
        _capi.check(self._lib.fr_zoom_path_zoom_to(self._h, C.byref(p), target_x, target_y, target_zoom, duration))
        self.zoom_animating, self.zoom_progress = True, 0.0

    def update_animation(self, delta_time: float) -> bool:
        """Advances the path; returns True where the reference recomputes its reference orbit (a keyframe was reached)."""
        p = self.state.to_params()
        anim, dirty, prog = C.c_int32(), C.c_int32(), C.c_float()
        _capi.check(self._lib.fr_zoom_path_update(self._h, delta_time, C.byref(p), C.byref(anim), C.byref(prog), C.byref(dirty)))
        self.state.center_x, self.state.center_y, self.state.zoom = p.center_x, p.center_y, p.zoom
        self.zoom_animating, self.zoom_progress = bool(anim.value), float(prog.value)
        return bool(dirty.value)
