"""Renderer: the Python face of the C-ABI render entry points.

Mirrors the reference's compute boundary for the hot path:
  ComputeEffectManager::dispatch(cmd, type, state, time, desc_set, extent)
      src/compute_effect_manager.h:435-468
  AnimationRenderer::RenderFrameCallback  bool(const FractalState&, width, height, path)
      src/animation_renderer.h:41-48
PyTorch is used only for device memory and streams; every pixel is computed by the HIP
kernels behind libfractalrenderer_amd.so.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _capi
from .state import FractalState, FractalType, Precision


@dataclass(frozen=True)
class Shard:
    """Row-strip sharding of one frame (fr_shard): strips of rows_per_strip rows dealt
    round-robin to nparts parts; this part renders only its own strips, packed."""
    part: int = 0
    nparts: int = 1
    rows_per_strip: int = 0

    def to_c(self) -> _capi.fr_shard:
        return _capi.fr_shard(self.part, self.nparts, self.rows_per_strip)

    def rows(self, height: int) -> int:
        s = self.to_c()
        return int(_capi.lib().fr_shard_rows(C.byref(s), height))

    def global_rows(self, height: int) -> np.ndarray:
        """frame row index of every packed local row of this part"""
        s = self.to_c()
        n = self.rows(height)
        L = _capi.lib()
        return np.array([L.fr_shard_global_row(C.byref(s), height, r) for r in range(n)], dtype=np.int64)


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


class Renderer:
    """One fr_ctx bound to one HIP device."""

    def __init__(self, device: int = 0, timing: bool = True):
        """timing: option "timing" -- the event pair around every render that last_kernel_ms() reads.  The C library's
        default is OFF (two timed event records cost a frame ~4.7 us); this wrapper, which tests and tools measure with,
        switches it on unless told otherwise (bench.py's timed contexts are told otherwise)."""
        self._lib = _capi.lib()
        h = C.c_void_p()
        _capi.check(self._lib.fr_ctx_create(int(device), C.byref(h)))
        self._ctx = h
        self.device = int(device)
        if timing:
            self.set_option("timing", 1)

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.fr_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def compute_units(self) -> int:
        return _capi.check(self._lib.fr_ctx_compute_units(self._ctx))

    def set_option(self, name: str, value: int) -> None:
        """fr_ctx_set_option (the public names), or fr_ctx_set_tuning for the internal queue / stream tuning names of
        csrc/fr_tuning.h (tests, tools, A/B measurements); 0 = automatic."""
        if name in _capi.TUNING_NAMES:
            _capi.check(self._lib.fr_ctx_set_tuning(self._ctx, name.encode(), int(value)))
        else:
            _capi.check(self._lib.fr_ctx_set_option(self._ctx, name.encode(), int(value)))

    def set_tuning(self, workgroups_per_cu: int = 0, subtiles_per_dequeue: int = 0, shape: int = 0,
                   run_min: int = 0, shift_bias: int = 0) -> None:
        """All queue knobs at once; 0 = automatic everywhere (see fr_ctx_set_option)."""
        for k, v in (("workgroups_per_cu", workgroups_per_cu), ("run_max", subtiles_per_dequeue),
                     ("subtile_shape", shape), ("run_min", run_min), ("shift_bias", shift_bias)):
            self.set_option(k, v)

    def reserve(self, state: FractalState, width: int, height: int, *, fractal_type: FractalType = FractalType.Mandelbrot,
                precision: Precision = Precision.F64, shard: Optional[Shard] = None) -> None:
        """fr_ctx_reserve: size the context's scratch for this geometry now, so that later sync=False renders of it
        (or of anything smaller) are launch-only."""
        p = state.to_params(fractal_type, precision)
        sh = shard.to_c() if shard else None
        _capi.check(self._lib.fr_ctx_reserve(self._ctx, C.byref(p), width, height, C.byref(sh) if sh is not None else None))

    def check(self) -> None:
        """fr_ctx_check: raises if a render that completed on this context lost pixels (after the caller's own sync)."""
        _capi.check(self._lib.fr_ctx_check(self._ctx))

    def last_grid(self) -> int:
        return _capi.check(self._lib.fr_ctx_last_grid(self._ctx)) & 0xFFFF

    def last_stages(self) -> int:
        return _capi.check(self._lib.fr_ctx_last_grid(self._ctx)) >> 16

    def last_kernel_ms(self) -> float:
        return float(self._lib.fr_ctx_last_kernel_ms(self._ctx))

    def last_pool_closing(self) -> int:
        """internal (fr_tuning.h): 1 / 0 = the lane pool of the most recent render looked / did not look for cycles, -1 = none"""
        return int(self._lib.fr_ctx_last_pool_closing(self._ctx))

    # -- plane plumbing ------------------------------------------------------------------
    @staticmethod
    def _ptr(x, want_dtype: str, nelem: int, what: str):
        if x is None:
            return None, None
        if _is_torch(x):
            if not x.is_contiguous():
                raise ValueError(f"{what}: tensor must be contiguous")
            if str(x.dtype).replace("torch.", "") != want_dtype:
                raise ValueError(f"{what}: dtype {x.dtype}, expected {want_dtype}")
            if x.numel() != nelem:
                raise ValueError(f"{what}: {x.numel()} elements, expected {nelem}")
            return x.data_ptr(), ("device" if x.is_cuda else "host")
        a = x
        if not isinstance(a, np.ndarray) or not a.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{what}: need a C-contiguous numpy array or torch tensor")
        if a.dtype != np.dtype(want_dtype):
            raise ValueError(f"{what}: dtype {a.dtype}, expected {want_dtype}")
        if a.size != nelem:
            raise ValueError(f"{what}: {a.size} elements, expected {nelem}")
        return a.ctypes.data, "host"

    def _output(self, precision: Precision, rows: int, width: int, rgba, nu, it) -> _capi.fr_output:
        # (also called with self = None by Node.render: uses nothing of the instance)
        npx = rows * width
        nu_dtype = "float64" if precision == Precision.F64 else "float32"   # Deep_Zoom callers pass Precision.F32
        kinds = set()
        o = _capi.fr_output()
        for name, x, dt, n in (("rgba", rgba, "float32", npx * 4), ("nu", nu, nu_dtype, npx), ("iter", it, "int32", npx)):
            p, kind = Renderer._ptr(x, dt, n, name)
            setattr(o, name, p)
            if kind:
                kinds.add(kind)
        if len(kinds) > 1:
            raise ValueError("output planes must all be host or all be device memory")
        o.memory = _capi.FR_MEM_DEVICE if kinds == {"device"} else _capi.FR_MEM_HOST
        return o

    # -- render ----------------------------------------------------------------------------
    def render(self, state: FractalState, width: int, height: int, *,
               fractal_type: FractalType = FractalType.Mandelbrot,
               precision: Precision = Precision.F64, post_chain: bool = False,
               rgba=None, nu=None, iter=None, shard: Optional[Shard] = None,
               stream: Optional[int] = None, sync: bool = True) -> None:
        """The reference's render(viewport, max_iter, out_buffer) surface.

        rgba: rows*W*4 float32, nu: rows*W float64 (F64) / float32 (F32), iter: rows*W int32;
        numpy arrays (host, PCIe-inclusive path) or torch CUDA tensors (device, no copies).
        sync=False enqueues on `stream` (a raw hipStream_t handle, e.g.
        torch.cuda.current_stream().cuda_stream) and returns at once (device planes only).
        """
        p = state.to_params(fractal_type, precision, post_chain)
        rows = shard.rows(height) if shard else height
        if rows == 0:
            _capi.check(self._lib.fr_params_validate(C.byref(p), width, height))
            return                                   # this part owns no rows of the frame
        out = self._output(precision, rows, width, rgba, nu, iter)
        sh = shard.to_c() if shard else None
        shp = C.byref(sh) if sh is not None else None
        if sync:
            if stream is not None:
                raise ValueError("stream is only meaningful with sync=False")
            _capi.check(self._lib.fr_render_shard(self._ctx, C.byref(p), width, height, shp, C.byref(out)))
        else:
            _capi.check(self._lib.fr_render_shard_async(self._ctx, C.byref(p), width, height, shp,
                                                        C.byref(out), C.c_void_p(stream or 0)))

    def dispatch(self, fractal_type: FractalType, state: FractalState, extent: tuple, **kw) -> None:
        """Name-for-name mirror of ComputeEffectManager::dispatch (type, state, extent);
        the Vulkan command buffer / descriptor set arguments become the output planes."""
        self.render(state, int(extent[0]), int(extent[1]), fractal_type=fractal_type, **kw)

    def render_frame(self, state: FractalState, width: int, height: int, path: str,
                     fractal_type: FractalType = FractalType.Mandelbrot,
                     precision: Precision = Precision.F32) -> bool:
        """The RenderFrameCallback itself: bool(const FractalState&, width, height, path)
        (src/animation_renderer.h:41-48): render, 8-bit export, PNG -- all behind fr_render_frame_png.
        Returns False on failure, as the reference's callback does."""
        p = state.to_params(fractal_type, precision)
        return self._lib.fr_render_frame_png(self._ctx, C.byref(p), width, height, os.fsencode(path)) == _capi.FR_OK

    @staticmethod
    def colorize_supported(state: FractalState, fractal_type: FractalType = FractalType.Mandelbrot,
                           precision: Precision = Precision.F64) -> bool:
        """fr_colorize_supported: the colour plane is a function of the smooth-count plane alone."""
        p = state.to_params(fractal_type, precision)
        return bool(_capi.lib().fr_colorize_supported(C.byref(p)))

    def colorize(self, state: FractalState, nu, rgba, *, fractal_type: FractalType = FractalType.Mandelbrot,
                 precision: Precision = Precision.F64, post_chain: bool = False,
                 stream: Optional[int] = None) -> None:
        """fr_colorize_async: device nu plane (float64 for F64, float32 for F32) -> device RGBA f32 plane,
        bit-identical to the rgba plane render() writes.  Enqueued on `stream` (raw hipStream_t), no sync."""
        p = state.to_params(fractal_type, precision, post_chain)
        n = nu.numel()
        p_nu, k1 = self._ptr(nu, "float64" if precision == Precision.F64 else "float32", n, "nu")
        p_rgba, k2 = self._ptr(rgba, "float32", n * 4, "rgba")
        if k1 != "device" or k2 != "device":
            raise ValueError("colorize needs device tensors")
        _capi.check(self._lib.fr_colorize_async(self._ctx, C.byref(p), n, p_nu, p_rgba, C.c_void_p(stream or 0)))

    def export_rgb16(self, rgba, width: int, height: int, out=None, through_half: bool = False, stream: Optional[int] = None):
        """16-bit export of export_print_quality (src/vk_engine.cpp:2054-2073): clamp, *65535, flip.
        stream (raw hipStream_t, device tensors only): enqueue there and return without waiting (fr_export_rgb16_async)."""
        p_in, kind_in = self._ptr(rgba, "float32", width * height * 4, "rgba")
        if out is None:
            if kind_in == "device":
                import torch
                out = torch.empty((height, width, 3), dtype=torch.int16, device=rgba.device)   # torch has no uint16 math: raw bits
            else:
                out = np.empty((height, width, 3), np.uint16)
        want = "int16" if _is_torch(out) else "uint16"
        p_out, kind_out = self._ptr(out, want, width * height * 3, "rgb16")
        if kind_in != kind_out:
            raise ValueError("rgba and rgb16 must live in the same memory kind")
        mem = _capi.FR_MEM_DEVICE if kind_in == "device" else _capi.FR_MEM_HOST
        if stream is not None:
            if kind_in != "device":
                raise ValueError("stream is only meaningful for device tensors")
            _capi.check(self._lib.fr_export_rgb16_async(self._ctx, p_in, width, height, p_out, int(through_half), C.c_void_p(stream)))
        else:
            _capi.check(self._lib.fr_export_rgb16(self._ctx, p_in, width, height, p_out, mem, int(through_half)))
        return out

    def export_rgb8(self, rgba, width: int, height: int, out=None, through_half: bool = False, stream: Optional[int] = None):
        """8-bit export of VulkanEngine::render_animation_frame (src/vk_engine.cpp:1344-1371):
        second ACES + gamma, u8 truncation, vertical flip.
        stream (raw hipStream_t, device tensors only): enqueue there and return without waiting (fr_export_rgb8_async)."""
        p_in, kind_in = self._ptr(rgba, "float32", width * height * 4, "rgba")
        if out is None:
            if kind_in == "device":
                import torch
                out = torch.empty((height, width, 3), dtype=torch.uint8, device=rgba.device)
            else:
                out = np.empty((height, width, 3), np.uint8)
        p_out, kind_out = self._ptr(out, "uint8", width * height * 3, "rgb8")
        if kind_in != kind_out:
            raise ValueError("rgba and rgb8 must live in the same memory kind")
        mem = _capi.FR_MEM_DEVICE if kind_in == "device" else _capi.FR_MEM_HOST
        if stream is not None:
            if kind_in != "device":
                raise ValueError("stream is only meaningful for device tensors")
            _capi.check(self._lib.fr_export_rgb8_async(self._ctx, p_in, width, height, p_out, int(through_half), C.c_void_p(stream)))
        else:
            _capi.check(self._lib.fr_export_rgb8(self._ctx, p_in, width, height, p_out, mem, int(through_half)))
        return out


class Node:
    """fr_node: one frame over the GPUs of a node behind the C ABI -- one process, one render context, stream and host
    worker thread per device, parts rendered concurrently, the frame assembled on devices[root] by in-place peer stores
    or by an RCCL gather (include/fractalrenderer_amd.h).  `devices` may repeat an ordinal (render lanes on one card)."""

    def __init__(self, devices):
        self._lib = _capi.lib()
        devs = [int(d) for d in devices]
        arr = (C.c_int * len(devs))(*devs)
        h = C.c_void_p()
        _capi.check(self._lib.fr_node_create(arr, len(devs), C.byref(h)))
        self._node = h
        self.devices = devs

    def close(self) -> None:
        if getattr(self, "_node", None):
            self._lib.fr_node_destroy(self._node)
            self._node = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, name: str, value: int) -> None:
        _capi.check(self._lib.fr_node_set_option(self._node, name.encode(), int(value)))

    def set_tuning(self, name: str, value: int) -> None:
        """fr_node_set_tuning (internal, csrc/fr_tuning.h): fault injection, the one-card RCCL loopback, context tuning."""
        _capi.check(self._lib.fr_node_set_tuning(self._node, name.encode(), int(value)))

    def render(self, state: FractalState, width: int, height: int, *, root: int = 0,
               fractal_type: FractalType = FractalType.Mandelbrot, precision: Precision = Precision.F64,
               post_chain: bool = False, rgba=None, nu=None, iter=None, sync: bool = True) -> None:
        """fr_node_render (sync=False: fr_node_render_async; call wait())."""
        p = state.to_params(fractal_type, precision, post_chain)
        out = Renderer._output(None, precision, height, width, rgba, nu, iter)
        fn = self._lib.fr_node_render if sync else self._lib.fr_node_render_async
        _capi.check(fn(self._node, C.byref(p), width, height, int(root), C.byref(out)))

    def submit(self, state: FractalState, width: int, height: int, *, root: int = 0,
               fractal_type: FractalType = FractalType.Mandelbrot, precision: Precision = Precision.F64,
               post_chain: bool = False, rgba=None, nu=None, iter=None) -> int:
        """fr_node_submit: hands one frame to the node and returns its ticket without waiting; up to "slots" frames are in
        flight.  The planes must stay alive until wait_frame(ticket) / wait()."""
        p = state.to_params(fractal_type, precision, post_chain)
        out = Renderer._output(None, precision, height, width, rgba, nu, iter)
        t = C.c_uint64(0)
        _capi.check(self._lib.fr_node_submit(self._node, C.byref(p), width, height, int(root), C.byref(out), C.byref(t)))
        return int(t.value)

    def wait_frame(self, ticket: int) -> None:
        _capi.check(self._lib.fr_node_wait_frame(self._node, int(ticket)))

    def wait(self) -> None:
        _capi.check(self._lib.fr_node_wait(self._node))

    def in_flight(self) -> int:
        return int(self._lib.fr_node_in_flight(self._node))

    def render_animation(self, anim, output_folder: Optional[str], *, base: Optional[FractalState] = None,
                         fractal_type: FractalType = FractalType.Mandelbrot, precision: Precision = Precision.F32,
                         width: int = 0, height: int = 0, first_frame: int = 0, frame_count: int = 0, frame_step: int = 0,
                         max_iterations: int = 0, on_frame_complete=None, raw_fd: int = 0) -> int:
        """fr_node_render_animation: AnimationRenderer::start_render (src/animation_renderer.cpp:26-152) over the node's
        devices, end to end (render -> 8-bit export on the frame's root -> PNG).  `anim`: an AnimationSystem.  Returns the
        number of frames written; on_frame_complete(frame, total) returning True cancels.  raw_fd > 0: packed RGB24 frames to
        that file descriptor (an encoder's stdin) instead of PNG files; output_folder may then be None."""
        p = (base if base is not None else anim.fractal_state).to_params(fractal_type, precision)
        cb = _capi.FR_FRAME_CALLBACK((lambda f, t, _u: 1 if on_frame_complete(f, t) else 0) if on_frame_complete else 0)
        o = _capi.fr_anim_render_options(int(width), int(height), int(first_frame), int(frame_count), int(frame_step),
                                         int(max_iterations), 0, cb, None, int(raw_fd), 0)
        n = C.c_int32(0)
        folder = os.fsencode(output_folder) if output_folder is not None else None
        _capi.check(self._lib.fr_node_render_animation(self._node, anim._h, C.byref(p), C.byref(o), folder, C.byref(n)))
        return int(n.value)

    def rccl_usable(self) -> bool:
        return bool(self._lib.fr_node_rccl_usable(self._node))

    def last_gather(self) -> int:
        return int(self._lib.fr_node_last_gather(self._node))

    def last_kernel_ms(self, part: int) -> float:
        return float(self._lib.fr_node_last_kernel_ms(self._node, int(part)))


def rccl_selftest(device: int = 0, nbytes: int = 1 << 20) -> int:
    """fr_node_rccl_selftest (internal): plugin load, one-rank communicator, a grouped ncclSend / ncclRecv pair on a stream,
    compared on the host.  Returns ncclGetVersion()."""
    v = C.c_int(0)
    _capi.check(_capi.lib().fr_node_rccl_selftest(int(device), int(nbytes), C.byref(v)))
    return int(v.value)


def mapped_runtimes() -> list:
    """fr_node_mapped_runtimes (internal): the librccl / libamdhip64 / libfractalrenderer_amd files this process has mapped."""
    buf = C.create_string_buffer(1 << 16)
    _capi.check(_capi.lib().fr_node_mapped_runtimes(buf, len(buf)))
    return [ln for ln in buf.value.decode().split("\n") if ln]


def export8_thresholds(host_powf: bool = False) -> np.ndarray:
    """fr_export8_thresholds (internal): t[b] = the smallest float32 a of [0, 1] with (uint8)(powf(a, 1/2.2f) * 255) >= b,
    b = 0..255, and t[256] = inf -- the table the 8-bit export kernel corrects its gamma estimate against: baked in, from the
    correctly rounded single-precision power (tools/gen_export8_table.py).  host_powf: the same by bisection with this host's libm."""
    t = (C.c_float * 257)()
    (_capi.lib().fr_export8_thresholds_host_powf if host_powf else _capi.lib().fr_export8_thresholds)(t)
    return np.frombuffer(t, dtype=np.float32).copy()


def write_png(path: str, rgb: np.ndarray, texts=None, print_metadata: bool = False) -> None:
    """fr_write_png: (H, W, 3) uint8 or uint16 -> PNG (8-bit: the animation frames, src/vk_engine.cpp:1374-1381;
    16-bit + print_metadata: the print export, src/vk_engine.cpp:2114-2208)."""
    a = np.ascontiguousarray(rgb)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError("rgb must be (H, W, 3) uint8 or uint16")
    items = list((texts or {}).items())
    arr = (_capi.fr_png_text * max(1, len(items)))()
    keep = []
    for k, (key, val) in enumerate(items):
        kb, vb = key.encode("latin-1"), val.encode("latin-1", "replace")
        keep += [kb, vb]
        arr[k].key, arr[k].text = kb, vb
    _capi.check(_capi.lib().fr_write_png(os.fsencode(path), a.shape[1], a.shape[0], 8 * a.dtype.itemsize,
                                         a.ctypes.data, arr, len(items), int(print_metadata)))


def write_raw_rgb24(fd: int, rgb8: np.ndarray) -> None:
    """fr_write_raw_rgb24: one packed RGB24 frame to a file descriptor (an encoder's stdin pipe)."""
    a = np.ascontiguousarray(rgb8, np.uint8)
    _capi.check(_capi.lib().fr_write_raw_rgb24(fd, a.ctypes.data, a.shape[1], a.shape[0]))


def frame_path(folder: str, frame: int) -> str:
    """frame_%06d.png naming of AnimationRenderer::start_render (src/animation_renderer.cpp:86-88)."""
    buf = C.create_string_buffer(4096)
    _capi.check(_capi.lib().fr_frame_path(os.fsencode(folder), frame, buf, len(buf)))
    return os.fsdecode(buf.value)
