"""Host-side logic behind the C ABI (no GPU, no compute calls): parameter block, push-constant
packing, validation, row-strip arithmetic, .franim I/O, keyframe interpolation, frame timing,
reference orbit, exported symbols, and the "no device -> loud failure" contract."""
import ctypes as C
import json
import math
import os
import re
import sys
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


# ---- C ABI surface -----------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol(fr):
    hdr = open(os.path.join(ROOT, "include", "fractalrenderer_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = C.CDLL(fr._capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert declared == set(fr._capi.SIGNATURES), declared ^ set(fr._capi.SIGNATURES)


def test_struct_layout_matches_header(fr):
    # fr_params: 2 i32, 3 f64, i32+f32, 2 f64, then 4-byte fields -> 112 bytes with natural alignment
    assert C.sizeof(fr._capi.fr_params) == 112
    assert fr._capi.fr_params.center_x.offset == 8 and fr._capi.fr_params.julia_c_real.offset == 40
    assert C.sizeof(fr._capi.fr_output) == 32 and C.sizeof(fr._capi.fr_shard) == 12
    major, minor = C.c_int(), C.c_int()
    fr.lib().fr_version(C.byref(major), C.byref(minor))
    assert (major.value, minor.value) == (1, 1)       # 1.1: frames in flight on a node


def test_ctypes_mirror_matches_the_compiled_header(fr, tmp_path):
    """sizeof / offsetof of the ABI's structs as a C compiler sees include/fractalrenderer_amd.h, against the ctypes mirror
    (_capi.py) field by field: a field added on one side only would shift everything behind it silently."""
    import shutil
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    structs = {"fr_params": fr._capi.fr_params, "fr_output": fr._capi.fr_output, "fr_shard": fr._capi.fr_shard,
               "fr_anim_render_options": fr._capi.fr_anim_render_options, "fr_keyframe": fr._capi.fr_keyframe}
    lines = ["#include <stdio.h>", "#include <stddef.h>", "#include \"fractalrenderer_amd.h\"", "int main(void) {"]
    for name, ct in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c11", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, ct in structs.items():
        assert int(got[name]) == C.sizeof(ct), name
        for fname, _ in ct._fields_:
            assert int(got[f"{name}.{fname}"]) == getattr(ct, fname).offset, (name, fname)


def test_raw_frame_writer_survives_a_dead_reader(fr):
    """fr_write_raw_rgb24 feeds an encoder's stdin.  Whole frames arrive byte for byte through a pipe (short writes retried);
    a reader that has gone away must come back as FR_ERR_IO -- not as the SIGPIPE that would end the host process (run in a
    child process with the default disposition, which Python itself does not have) -- and leave no signal pending."""
    code = r"""
import ctypes as C, os, signal, sys, threading
sys.path.insert(0, %r)
import numpy as np
import fractalrenderer_amd as fr
signal.signal(signal.SIGPIPE, signal.SIG_DFL)
L = fr.lib()
L.fr_write_raw_rgb24.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32]
W, H = 640, 400
frame = (np.arange(W * H * 3, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8)
rd, wr = os.pipe()
got = bytearray()
def reader():
    with os.fdopen(rd, "rb") as f:
        while True:
            b = f.read(4096)
            if not b: break
            got.extend(b)
t = threading.Thread(target=reader); t.start()
assert L.fr_write_raw_rgb24(wr, frame.ctypes.data, W, H) == 0
os.close(wr); t.join()
assert bytes(got) == frame.tobytes()
rd, wr = os.pipe()
os.close(rd)                                   # the encoder died
st = L.fr_write_raw_rgb24(wr, frame.ctypes.data, W, H)
assert st == fr._capi.FR_ERR_IO, st
assert signal.SIGPIPE not in signal.sigpending()
st = L.fr_write_raw_rgb24(wr, frame.ctypes.data, W, H)       # and again
assert st == fr._capi.FR_ERR_IO, st
os.close(wr)
print("alive")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("alive"), (out.returncode, out.stderr[-2000:])


def test_no_device_fails_loudly(fr):
    """The product path has no CPU fallback: without a GPU the context cannot be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(fr.FractalRendererError) as e:
        fr.Renderer(0)
    assert e.value.status == fr._capi.FR_ERR_NO_DEVICE and "no CPU path" in str(e.value)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "fractalrenderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle" not in text.lower(), f"{fn} mentions the oracle"


# ---- defaults / reset (src/fractal_state.h) ---------------------------------------------------------------
def test_params_default_and_reset(fr):
    p = fr._capi.fr_params()
    assert fr.lib().fr_params_default(C.byref(p)) == 0
    assert (p.center_x, p.center_y, p.zoom, p.max_iterations) == (-0.5, 0.0, 3.0, 256)
    assert p.bailout == 4.0 and p.antialiasing_samples == 1 and p.palette_mode == 0
    assert p.julia_c_real == float(f32(-0.7)) and p.julia_c_imag == float(f32(0.27015))
    assert (p.color_offset, p.color_scale, p.orbit_trap_radius, p.stripe_density) == (0.0, 1.0, 0.5, 10.0)
    assert (p.color_brightness, p.color_saturation, p.color_contrast) == (1.0, 1.0, 1.0)
    assert p.interior_style == 0 and p.orbit_trap_enabled == 0 and p.stripe_enabled == 0 and p.flags == 0
    p.zoom, p.center_x, p.max_iterations, p.palette_mode = 0.01, 0.3, 999, 4
    fr.lib().fr_params_reset(C.byref(p))
    assert (p.center_x, p.center_y, p.zoom, p.max_iterations) == (-0.5, 0.0, 1.5, 256)    # reset() zoom is 1.5
    assert p.palette_mode == 4                                                              # untouched by reset()
    s = fr.FractalState()
    q = s.to_params()
    d = fr._capi.fr_params(); fr.lib().fr_params_default(C.byref(d))
    assert bytes(q) == bytes(d)


# ---- push constants (src/compute_effect_manager.h:84-140) -----------------------------------------------------
def test_push_constants_mandelbrot(fr, oracle):
    s = fr.FractalState(center_x=-0.743643887037151, center_y=0.13182590420533, zoom=1e-6, max_iterations=16384,
                        color_offset=0.25, color_scale=2.0, bailout=8.0, palette_mode=3, antialiasing_samples=2,
                        interior_style=2, orbit_trap_enabled=True, orbit_trap_radius=0.75, stripe_density=12.5,
                        stripe_enabled=True, color_brightness=1.1, color_saturation=0.9, color_contrast=1.2)
    pc = fr.pack_push_constants(s, fr.FractalType.Mandelbrot)
    expect = [f32(-0.743643887037151), f32(0.13182590420533), f32(1e-6), 16384.0,      # data1
              0.25, 2.0, 8.0, 3.0,                                                      # data2
              2.0, 2.0, 1.0, 0.75,                                                      # data3
              12.5, 1.0, f32(1.1), f32(0.9),                                            # data4
              f32(1.2), 0.0, 0.0, 0.0]                                                  # data5
    assert pc.dtype == np.float32 and pc.tobytes() == np.array(expect, np.float32).tobytes()
    # the double -> float narrowing is the reference's precision ceiling: centre loses its low bits
    assert float(pc[0]) != s.center_x
    op = oracle.OracleParams(center_x=s.center_x, center_y=s.center_y, zoom=s.zoom, max_iterations=16384,
                             color_offset=0.25, color_scale=2.0, bailout=8.0, palette_mode=3, aa=2, interior_style=2,
                             orbit_trap_enabled=1, orbit_trap_radius=0.75, stripe_density=12.5, stripe_enabled=1,
                             brightness=1.1, saturation=0.9, contrast=1.2)
    assert oracle.pack_push_constants(op).tobytes() == pc.tobytes()


def test_push_constants_julia(fr, oracle):
    s = fr.FractalState(center_x=0.0, zoom=3.0, max_iterations=2048, julia_c_real=-0.8, julia_c_imag=0.156,
                        color_offset=0.1, color_scale=1.5, palette_mode=7, antialiasing_samples=4)
    pc = fr.pack_push_constants(s, fr.FractalType.JuliaSet)
    expect = [0.0, 0.0, 3.0, 2048.0, f32(-0.8), f32(0.156), 4.0, f32(0.1),
              4.0, 1.5, 1.0, 1.0, 1.0, 7.0, 0.0, 0.0, 0, 0, 0, 0]
    assert pc.tobytes() == np.array(expect, np.float32).tobytes()
    op = oracle.OracleParams(fractal=1, center_x=0.0, max_iterations=2048, julia_c_real=-0.8, julia_c_imag=0.156,
                             color_offset=0.1, color_scale=1.5, palette_mode=7, aa=4)
    assert oracle.pack_push_constants(op).tobytes() == pc.tobytes()
    with pytest.raises(fr.FractalRendererError) as e:
        fr.pack_push_constants(s, fr.FractalType.Phoenix)
    assert e.value.status == fr._capi.FR_ERR_UNSUPPORTED


def test_push_constants_burning_ship(fr, oracle):
    """src/compute_effect_manager.h:142-171: the Mandelbrot layout."""
    s = fr.FractalState(center_x=-1.755, center_y=-0.03, zoom=0.08, max_iterations=1024, color_offset=0.2,
                        color_scale=2.0, palette_mode=9, antialiasing_samples=2, interior_style=3,
                        orbit_trap_enabled=True, orbit_trap_radius=0.6, stripe_enabled=True, stripe_density=7.0,
                        color_brightness=1.1, color_saturation=0.9, color_contrast=1.2)
    pc = fr.pack_push_constants(s, fr.FractalType.BurningShip)
    assert pc.tobytes() == fr.pack_push_constants(s, fr.FractalType.Mandelbrot).tobytes()
    expect = [f32(-1.755), f32(-0.03), f32(0.08), 1024.0, f32(0.2), 2.0, 4.0, 9.0,
              2.0, 3.0, 1.0, f32(0.6), 7.0, 1.0, f32(1.1), f32(0.9), f32(1.2), 0, 0, 0]
    assert pc.tobytes() == np.array(expect, np.float32).tobytes()
    op = oracle.OracleParams(fractal=2, center_x=-1.755, center_y=-0.03, zoom=0.08, max_iterations=1024,
                             color_offset=0.2, color_scale=2.0, palette_mode=9, aa=2, interior_style=3,
                             orbit_trap_enabled=1, orbit_trap_radius=0.6, stripe_enabled=1, stripe_density=7.0,
                             brightness=1.1, saturation=0.9, contrast=1.2)
    assert oracle.pack_push_constants(op).tobytes() == pc.tobytes()


def test_push_constants_deep_zoom(fr, oracle):
    """src/compute_effect_manager.h:236-324: float-float split of centre/zoom, reference_iterations in data4.y"""
    for kw in (dict(center_x=-0.743643887037151, center_y=0.13182590420533, zoom=1e-6, max_iterations=2000, use_perturbation=True),
               dict(center_x=-0.75, center_y=0.1, zoom=100.0, max_iterations=300, use_perturbation=True, palette_mode=2,
                    color_offset=0.25, color_scale=3.0, bailout=8.0, antialiasing_samples=2),
               dict(center_x=0.3, center_y=0.5, zoom=1e-3, max_iterations=100, use_perturbation=False)):
        s = fr.FractalState(**kw)
        pc = fr.pack_push_constants(s, fr.FractalType.Deep_Zoom)
        hi = f32(s.center_x)
        assert pc[0] == hi and pc[1] == f32(s.center_x - float(hi))
        assert pc[6] == s.max_iterations and pc[7] == float(s.use_perturbation)
        n = len(oracle.reference_orbit(s.center_x, s.center_y, s.max_iterations)) if s.use_perturbation else 0
        assert pc[13] == n and pc[12] == s.antialiasing_samples and pc[15] == 3.0
        op = oracle.OracleParams(fractal=5, precision=0, center_x=s.center_x, center_y=s.center_y, zoom=s.zoom,
                                 max_iterations=s.max_iterations, use_perturbation=int(s.use_perturbation),
                                 palette_mode=s.palette_mode, color_offset=s.color_offset, color_scale=s.color_scale,
                                 bailout=s.bailout, aa=s.antialiasing_samples)
        assert oracle.pack_push_constants(op).tobytes() == pc.tobytes()


# ---- validation ---------------------------------------------------------------------------------------------
def test_validation(fr):
    L = fr.lib()

    def st(w=64, h=64, **kw):
        p = fr.FractalState().to_params()
        for k, v in kw.items():
            setattr(p, k, v)
        return L.fr_params_validate(C.byref(p), w, h)

    assert st() == 0
    assert st(w=0) == fr._capi.FR_ERR_INVALID_ARG and st(h=0) == fr._capi.FR_ERR_INVALID_ARG
    assert st(w=65536, h=32768) == fr._capi.FR_ERR_INVALID_ARG          # 2^31 pixels
    assert st(w=32768, h=32768) == 0
    assert st(max_iterations=0) == fr._capi.FR_ERR_INVALID_ARG
    assert st(max_iterations=(1 << 24) + 1) == fr._capi.FR_ERR_INVALID_ARG and st(max_iterations=1 << 24) == 0
    assert st(zoom=0.0) == fr._capi.FR_ERR_INVALID_ARG and st(zoom=math.inf) == fr._capi.FR_ERR_INVALID_ARG
    assert st(zoom=-2.0) == 0                                            # a mirrored view is legal
    assert st(center_x=math.nan) == fr._capi.FR_ERR_INVALID_ARG
    assert st(bailout=0.0) == fr._capi.FR_ERR_INVALID_ARG and st(bailout=math.inf) == fr._capi.FR_ERR_INVALID_ARG
    assert st(antialiasing_samples=17) == fr._capi.FR_ERR_INVALID_ARG and st(antialiasing_samples=0) == 0
    assert st(precision=2) == fr._capi.FR_ERR_INVALID_ARG
    for t in (fr.FractalType.Mandelbulb, fr.FractalType.Phoenix):
        assert st(fractal_type=int(t)) == fr._capi.FR_ERR_UNSUPPORTED
    assert st(fractal_type=int(fr.FractalType.Deep_Zoom)) == 0 and st(fractal_type=int(fr.FractalType.BurningShip)) == 0
    assert st(fractal_type=9) == fr._capi.FR_ERR_INVALID_ARG
    assert b"outside the hot path" in L.fr_last_error() or b"unknown" in L.fr_last_error()
    assert L.fr_status_string(-4) == b"fractal type outside the hot path"


# ---- row strips ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,nparts,R", [(8192, 8, 64), (4096, 8, 32), (1000, 8, 64), (23, 3, 5), (7, 8, 1),
                                        (5, 8, 64), (4096, 1, 0), (130, 4, 33), (1, 2, 1)])
def test_shard_partition(fr, H, nparts, R):
    owners = np.full(H, -1)
    total = 0
    for part in range(nparts):
        sh = fr.Shard(part, nparts, R)
        rows = sh.global_rows(H)
        assert len(rows) == sh.rows(H)
        assert np.all(np.diff(rows) > 0)                       # packed rows keep frame order
        assert np.all(owners[rows] == -1)                      # disjoint
        owners[rows] = part
        total += len(rows)
        RR = R if R else H
        assert np.all((rows // RR) % nparts == part)           # strip s belongs to part s % nparts
    assert total == H and np.all(owners >= 0)                  # exhaustive
    s = fr.Shard(0, nparts, R).to_c()
    assert fr.lib().fr_shard_global_row(C.byref(s), H, H + 5) == 0xFFFFFFFF


# ---- .franim (src/animation_system.cpp:221-313) -----------------------------------------------------------------
def py_interpolate(kfs, duration, time):
    """Independent restatement of AnimationSystem::interpolate (src/animation_system.cpp:82-181)
    in numpy scalar arithmetic (float32 where the reference uses float)."""
    time = f32(min(max(f32(time), f32(0.0)), f32(duration)))
    k1, k2 = len(kfs) - 2, len(kfs) - 1
    for i in range(len(kfs) - 1):
        if time >= f32(kfs[i]["time"]) and time <= f32(kfs[i + 1]["time"]):
            k1, k2 = i, i + 1
            break
    a, b = kfs[k1], kfs[k2]
    td = f32(b["time"]) - f32(a["time"])
    if td < f32(0.001):
        return dict(center_x=a["center_x"], center_y=a["center_y"], zoom=a["zoom"],
                    max_iterations=a["max_iterations"], palette_mode=a["palette_mode"],
                    color_offset=f32(a["color_offset"]), color_scale=f32(a["color_scale"]))
    t = f32((time - f32(a["time"])) / td)
    it = b["interp_type"]
    if it == 1:
        t = f32(2.0) * t * t if t < f32(0.5) else f32(1.0) - f32(np.power(f32(-2.0) * t + f32(2.0), f32(2.0))) / f32(2.0)
    elif it in (2, 4):
        t = t * t
    elif it == 3:
        t = f32(1.0) - (f32(1.0) - t) * (f32(1.0) - t)
    t = f32(t)
    td_ = float(t)                                        # float t widened for the double expressions
    cx = a["center_x"] + td_ * (b["center_x"] - a["center_x"])
    cy = a["center_y"] + td_ * (b["center_y"] - a["center_y"])
    if a["zoom"] > 0 and b["zoom"] > 0:
        zoom = math.exp(math.log(a["zoom"]) + td_ * (math.log(b["zoom"]) - math.log(a["zoom"])))
    else:
        zoom = a["zoom"] + td_ * (b["zoom"] - a["zoom"])
    zoom = max(0.000001, zoom)
    iter_t = f32(0.0) if t < f32(0.33) else (f32(0.5) if t < f32(0.67) else f32(1.0))
    mi = int(f32(a["max_iterations"]) + iter_t * f32(b["max_iterations"] - a["max_iterations"]))
    return dict(center_x=cx, center_y=cy, zoom=zoom, max_iterations=mi,
                palette_mode=a["palette_mode"] if t < f32(0.5) else b["palette_mode"],
                color_offset=f32(a["color_offset"]) + t * (f32(b["color_offset"]) - f32(a["color_offset"])),
                color_scale=f32(a["color_scale"]) + t * (f32(b["color_scale"]) - f32(a["color_scale"])))


def test_franim_load_reference_sample(fr, golden):
    a = fr.AnimationSystem()
    assert a.load_from_file(golden["franim"])
    raw = json.load(open(golden["franim"]))
    i = a.info
    assert (i.duration, i.loop, i.target_fps, i.export_width, i.export_height) == (20.0, 1, 120, 2560, 1440)
    assert i.keyframe_count == 6 and a.name == "" and a.description == ""
    for kf, src in zip(a.get_keyframes(), raw["keyframes"]):
        assert kf.time == src["time"] and int(kf.interp_type) == src["interp_type"]
        for key in ("center_x", "center_y", "zoom", "max_iterations", "palette_mode", "color_offset", "color_scale"):
            assert getattr(kf.state, key) == src[key], key
        # fields the loader does not read stay at FractalState defaults (src/animation_system.cpp:290-298)
        assert kf.state.bailout == 4.0 and kf.state.antialiasing_samples == 1 and kf.state.julia_c_real == float(f32(-0.7))
    # frame arithmetic, src/animation_renderer.cpp:48,80
    assert a.frame_count() == 2400 and a.frame_time(600) == 5.0 and a.frame_time(1) == float(f32(1) / f32(120))


def test_franim_reference_recorded_keyframe(fr, golden):
    """The sample's 6th keyframe (t = 20.0) is the reference's own interpolate(20.0) result saved
    back as a keyframe: centre_y = 9.99999999995449e-06 and zoom = 0.0005000000000000001 are what
    key3 -> key4 evaluate to at t = 1 in double arithmetic.  A known answer produced BY the reference."""
    a = fr.AnimationSystem()
    assert a.load_from_file(golden["franim"])
    s = a.interpolate(20.0)
    last = json.load(open(golden["franim"]))["keyframes"][5]
    assert s.center_x == last["center_x"] == -1.7497
    assert s.center_y == last["center_y"] == 9.99999999995449e-06
    assert s.zoom == last["zoom"] == 0.0005000000000000001
    assert s.max_iterations == last["max_iterations"] == 1024


def test_franim_interpolate_matches_restatement(fr, golden):
    a = fr.AnimationSystem()
    assert a.load_from_file(golden["franim"])
    raw = json.load(open(golden["franim"]))
    times = [a.frame_time(f) for f in range(0, 2400, 37)] + [0.0, 4.999, 5.0, 5.001, 19.999, 20.0, 25.0, -1.0]
    for t in times:
        got = a.interpolate(t)
        exp = py_interpolate(raw["keyframes"], raw["duration"], t)
        for k, v in exp.items():
            assert getattr(got, k) == (float(v) if isinstance(v, np.floating) else v), (t, k)
        assert got.zoom >= 1e-6                                       # :145 floor


def test_franim_interp_types_and_edge_cases(fr):
    a = fr.AnimationSystem(fr.FractalState(zoom=7.0))
    assert a.interpolate(1.0).zoom == 7.0                           # no keyframes -> live state (:83)
    k0 = fr.FractalState(center_x=0.0, zoom=1.0, max_iterations=100, palette_mode=1, color_offset=0.0, bailout=8.0,
                         antialiasing_samples=2, orbit_trap_enabled=True, orbit_trap_radius=0.25,
                         julia_c_real=0.5, interior_style=1, stripe_enabled=True)
    a.add_keyframe(0.0, k0, fr.InterpolationType.Linear)
    assert a.interpolate(3.0).bailout == 8.0                        # one keyframe -> its state (:84)
    for it in fr.InterpolationType:
        b = fr.AnimationSystem()
        k1 = fr.FractalState(center_x=2.0, zoom=1e-4, max_iterations=1000, palette_mode=5, color_offset=1.0)
        b.add_keyframe(4.0, k1, it)                                  # added out of order: sorted by time (:16-17)
        b.add_keyframe(0.0, k0, fr.InterpolationType.Linear)
        assert [k.time for k in b.get_keyframes()] == [0.0, 4.0]
        s = b.interpolate(1.0)
        t = f32(0.25)
        e = {0: t, 1: f32(2) * t * t, 2: t * t, 3: f32(1) - (f32(1) - t) * (f32(1) - t), 4: t * t}[int(it)]
        assert s.center_x == float(e) * 2.0                          # easing of the SECOND keyframe (:107)
        assert s.zoom == pytest.approx(math.exp(float(e) * math.log(1e-4)), rel=1e-15)
        assert s.max_iterations == (100 if e < f32(0.33) else 550 if e < f32(0.67) else 1000)
        assert s.palette_mode == (1 if e < f32(0.5) else 5)
        # taken from key1 (:175-178); everything else reverts to FractalState defaults (:125)
        assert s.bailout == 8.0 and s.antialiasing_samples == 2 and s.orbit_trap_enabled and s.orbit_trap_radius == 0.25
        assert s.julia_c_real == float(f32(-0.7)) and s.interior_style == 0 and not s.stripe_enabled
    # duration grows to time + 1 when a keyframe lands beyond it (:20-22)
    c = fr.AnimationSystem()
    assert c.get_duration() == 10.0
    c.add_keyframe(12.0, k0)
    assert c.get_duration() == 13.0
    # negative/zero zoom -> linear interpolation, then the 1e-6 floor (:139-145)
    d = fr.AnimationSystem()
    d.add_keyframe(0.0, fr.FractalState(zoom=-1.0), fr.InterpolationType.Linear)
    d.add_keyframe(2.0, fr.FractalState(zoom=1.0), fr.InterpolationType.Linear)
    assert d.interpolate(0.5).zoom == 1e-6 and d.interpolate(1.5).zoom == 0.5


def test_franim_save_load_round_trip(fr, golden, tmp_path):
    a = fr.AnimationSystem()
    assert a.load_from_file(golden["franim"])
    out = str(tmp_path / "rt.franim")
    assert a.save_to_file(out)
    j = json.load(open(out))                                         # valid JSON
    assert len(j["keyframes"][0]) == 19                              # the writer's 19 keys (:235-255)
    assert list(j.keys()) == sorted(j.keys()) and list(j["keyframes"][0]) == sorted(j["keyframes"][0])
    b = fr.AnimationSystem()
    assert b.load_from_file(out)
    assert bytes(a.info) == bytes(b.info)
    for x, y in zip(a.get_keyframes(), b.get_keyframes()):
        assert x == y                                                # doubles survive the text round trip exactly
    assert not b.load_from_file(str(tmp_path / "missing.franim"))    # reference: returns false


def test_franim_parse_errors(fr, golden):
    good = json.load(open(golden["franim"]))
    a = fr.AnimationSystem()
    for key in ("name", "duration", "loop", "target_fps", "keyframes"):
        bad = dict(good); del bad[key]
        with pytest.raises(fr.FractalRendererError) as e:
            a.loads(json.dumps(bad))
        assert e.value.status == fr._capi.FR_ERR_PARSE and key in str(e.value)
    for key in ("time", "interp_type", "center_x", "zoom", "max_iterations", "palette_mode", "color_offset", "color_scale"):
        bad = json.loads(json.dumps(good)); del bad["keyframes"][2][key]
        with pytest.raises(fr.FractalRendererError):
            a.loads(json.dumps(bad))
    for text in ("", "{", "[1,2]", '{"name": "x",}', '{"name": "\\q"}', "nul"):
        with pytest.raises(fr.FractalRendererError):
            a.loads(text)
    # optional writer keys are honoured; integers given as floats are truncated like nlohmann's get<int>()
    g = json.loads(json.dumps(good))
    g["keyframes"][0].update(bailout=6.5, antialiasing_samples=2.9, orbit_trap_enabled=True, color_brightness=1.25,
                             max_iterations=300.7, rotation_y=1.0)
    g["name"] = "zürich \"q\"\n"
    a.loads(json.dumps(g))
    k = a.get_keyframes()[0].state
    assert (k.bailout, k.antialiasing_samples, k.orbit_trap_enabled, k.color_brightness, k.max_iterations) == (6.5, 2, True, 1.25, 300)
    assert a.name == g["name"]


def test_animation_renderer_frame_loop(fr, golden):
    a = fr.AnimationSystem()
    assert a.load_from_file(golden["franim"])
    calls = []

    def cb(state, w, h, path):
        calls.append((state, w, h, path))
        return len(calls) < 4                                       # 4th frame "fails"
    r = fr.AnimationRenderer(cb)
    assert r.start_render(a, "out", frames=range(0, 2400, 1000)) is True
    assert [c[3] for c in calls] == [os.path.join("out", "frame_%06d.png" % f) for f in (0, 1000, 2000)]
    assert calls[0][1:3] == (2560, 1440)                            # export size of the animation
    assert calls[1][0] == a.interpolate(a.frame_time(1000))
    assert r.start_render(a, "out", frames=range(5)) is False       # callback failure aborts (:109-116)
    assert fr.AnimationRenderer(None).start_render(a) is False      # no callback set (:207-210)
    one = fr.AnimationSystem(); one.add_keyframe(0.0, fr.FractalState())
    assert fr.AnimationRenderer(cb).start_render(one) is False      # needs two keyframes (:35-42)


# ---- reference orbit (src/deep_zoom_system.cpp:378-424) ---------------------------------------------------------------
def test_reference_orbit_matches_oracle(fr, oracle):
    L = fr.lib()
    for cx, cy, n in [(-0.5, 0.0, 300), (1.0, 0.0, 100), (0.0, 1.0, 50), (-0.743643887037151, 0.13182590420533, 5000),
                      (0.3, 0.5, 1000), (-2.0, 0.0, 10), (2.5, 0.0, 10), (1e200, 0.0, 10)]:
        buf = np.zeros((n, 2)); ln = C.c_int32()
        assert L.fr_reference_orbit(cx, cy, n, buf.ctypes.data, C.byref(ln)) == 0
        ref = oracle.reference_orbit(cx, cy, n)
        assert ln.value == len(ref) and np.array_equal(buf[:ln.value], ref)
    assert L.fr_reference_orbit(0.0, 0.0, 0, buf.ctypes.data, C.byref(ln)) == fr._capi.FR_ERR_INVALID_ARG


# ---- frame output (src/vk_engine.cpp:1374-1381, 2114-2208; src/animation_renderer.cpp:86-88) -----------------------
def test_write_png_8_and_16_bit(fr, tmp_path):
    import struct
    from pngdec import read_png
    rng = np.random.default_rng(3)
    a8 = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    p8 = str(tmp_path / "a8.png")
    fr.write_png(p8, a8)
    px, chunks = read_png(p8)
    assert np.array_equal(px, a8) and [c[0] for c in chunks] == ["IHDR", "IDAT", "IDAT", "IEND"]   # one band + the Adler-32
    a16 = rng.integers(0, 65536, (19, 31, 3), dtype=np.uint16)
    p16 = str(tmp_path / "a16.png")
    fr.write_png(p16, a16, texts={"Software": "fractalrenderer_amd", "Center": "(-0.5, 0.0)", "Iterations": "1024"},
                 print_metadata=True)
    px, chunks = read_png(p16)
    assert np.array_equal(px, a16)
    names = [c[0] for c in chunks]
    assert names == ["IHDR", "gAMA", "sRGB", "pHYs", "tEXt", "tEXt", "tEXt", "tIME", "IDAT", "IDAT", "IEND"]
    d = dict(chunks[:4])
    assert struct.unpack(">I", d["gAMA"])[0] == 45455                       # 1/2.2 (src/vk_engine.cpp:2145)
    assert d["sRGB"] == b"\x00"                                             # perceptual intent (:2146)
    assert struct.unpack(">IIB", d["pHYs"]) == (11811, 11811, 1)            # 300 dpi in pixels per metre (:2149-2152)
    texts = [c[1] for c in chunks if c[0] == "tEXt"]
    assert texts[0] == b"Software\x00fractalrenderer_amd" and texts[2] == b"Iterations\x001024"
    with pytest.raises(fr.FractalRendererError):
        fr.write_png(str(tmp_path / "no" / "dir.png"), a8)
    with pytest.raises(ValueError):
        fr.write_png(p8, a8[..., :2])


def test_write_png_parallel_bands(fr, tmp_path, monkeypatch):
    """Bands of ~1 MiB are deflated by worker threads and concatenated into one zlib stream (one IDAT chunk per
    band + one for the Adler-32); the partition depends on the image size only, so the bytes do not depend on
    the number of threads."""
    from pngdec import read_png
    rng = np.random.default_rng(5)
    # smooth-ish content (compressible) with noise, 5 bands of 499 rows + a partial one
    y, x = np.mgrid[0:2600, 0:700]
    img = np.stack([(x + y) % 256, (x * 3 + rng.integers(0, 8, x.shape)) % 256, (y // 3) % 256], axis=-1).astype(np.uint8)
    paths = []
    for threads in ("1", "3", "16"):
        monkeypatch.setenv("FR_PNG_THREADS", threads)
        p = str(tmp_path / ("t%s.png" % threads))
        fr.write_png(p, img)
        paths.append(p)
    blobs = [open(p, "rb").read() for p in paths]
    assert blobs[0] == blobs[1] == blobs[2]
    px, chunks = read_png(paths[0])
    assert np.array_equal(px, img)
    assert [c[0] for c in chunks].count("IDAT") == 6 + 1 and len(chunks[-2][1]) == 4
    assert len(blobs[0]) < img.nbytes * 4 // 5                                # it does compress (one channel is noisy)
    wide = rng.integers(0, 65536, (3, 200000, 3), dtype=np.uint16)            # rows longer than a band: one row per band
    pw = str(tmp_path / "wide.png")
    fr.write_png(pw, wide)
    px, chunks = read_png(pw)
    assert np.array_equal(px, wide) and [c[0] for c in chunks].count("IDAT") == 3 + 1


def test_frame_naming_and_raw_pipe(fr):
    assert fr.frame_path("animation_frames", 0) == "animation_frames/frame_000000.png"
    assert fr.frame_path("out", 123456) == "out/frame_123456.png"
    r, w = os.pipe()
    frame = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    fr.write_raw_rgb24(w, frame)
    os.close(w)
    got = os.read(r, 1000)
    os.close(r)
    assert got == frame.tobytes()
    with pytest.raises(fr.FractalRendererError):
        fr.write_raw_rgb24(-1, frame)


def build_c_client(tmp_path, source="client.c"):
    """gcc-compiles tests/c_client/<source> against include/fractalrenderer_amd.h and the built library."""
    import subprocess
    exe = str(tmp_path / ("fr_" + source[:-2]))
    libdir = os.path.join(ROOT, "fractalrenderer_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_client", source), "-o", exe,
           "-L" + libdir, "-lfractalrenderer_amd", "-lm", "-Wl,-rpath," + libdir]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_plain_c_client_host_entry_points(fr, golden, tmp_path):
    """The boundary is a C ABI: a C11 program that only includes the public header links against the library and
    drives the host-side entry points (params, push constants, shard arithmetic, .franim, PNG, naming)."""
    import subprocess
    exe = build_c_client(tmp_path)
    out = subprocess.run([exe, "host", golden["franim"], str(tmp_path / "c.png")], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "host ok", out.stderr


def test_node_entry_points_without_a_device(fr, tmp_path):
    """The multi-GPU entry points are part of the C ABI: the plain-C node client compiles against the public header and
    links; without a device fr_node_create fails loudly (no CPU path); the RCCL leg is its own library, which alone links
    librccl and exports what fr_node.cpp binds."""
    import subprocess
    exe = build_c_client(tmp_path, "node_client.c")
    assert os.path.exists(exe)
    devs = (C.c_int * 2)(0, 0)
    h = C.c_void_p()
    st = fr.lib().fr_node_create(devs, 2, C.byref(h))
    assert st == fr._capi.FR_ERR_NO_DEVICE and not h.value
    assert fr.lib().fr_node_create(devs, 0, C.byref(h)) == fr._capi.FR_ERR_INVALID_ARG
    libdir = os.path.join(ROOT, "fractalrenderer_amd")
    plugin = os.path.join(libdir, "libfractalrenderer_amd_rccl.so")
    assert os.path.exists(plugin)
    syms = subprocess.run(["nm", "-D", "--defined-only", plugin], capture_output=True, text=True).stdout
    for name in ("fr_rccl_init", "fr_rccl_destroy", "fr_rccl_group_start", "fr_rccl_group_end", "fr_rccl_send", "fr_rccl_recv",
                 "fr_rccl_version", "fr_rccl_abort"):
        assert re.search(r"\bT %s\b" % name, syms), name
    needed = subprocess.run(["objdump", "-p", plugin], capture_output=True, text=True).stdout
    assert "librccl.so" in needed
    main_needed = subprocess.run(["objdump", "-p", os.path.join(libdir, "libfractalrenderer_amd.so")], capture_output=True, text=True).stdout
    assert "librccl" not in main_needed          # single-GPU callers never map the 570 MB library


def test_export8_thresholds_against_an_exhaustive_scan(fr, oracle):
    """The 8-bit export kernel corrects its gamma estimate against 255 thresholds t[b] = the smallest float of [0, 1]
    whose byte (uint8)(powf(a, 1/2.2f) * 255) is >= b.  Since round 4 the table is baked into the library, generated from the
    CORRECTLY ROUNDED single-precision power (tools/gen_export8_table.py, mpmath at 200 bits) so that it does not depend on a
    host's libm.  That is only a description of the byte if the byte is monotone in a: the checker scans EVERY float of
    [0, 1] (1 065 353 217 of them) with the restated expression of src/vk_engine.cpp:1367-1368 -- its power being the
    double-precision pow rounded to float, an independent route to the same definition --, requires zero decreases and the
    same thresholds.  The deployment host's own powf is compared too and may differ by a float at a few thresholds (glibc's
    does at byte 33): reported, and bounded."""
    bad, first = oracle.export8_scan()
    assert bad == 0
    assert first[0] == 0 and (first != 0xFFFFFFFF).all()            # every byte value occurs
    assert (np.diff(first.astype(np.int64)) > 0).all()
    t = fr.export8_thresholds()
    assert t.shape == (257,) and t[0] == 0.0 and np.isinf(t[256])
    assert np.array_equal(t[:256].view(np.uint32), first)
    # the committed table is what the generator prints (mpmath is test infrastructure only)
    import subprocess
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_export8_table.py")], capture_output=True, text=True)
    assert gen.returncode == 0, gen.stderr
    assert gen.stdout == open(os.path.join(ROOT, "fractalrenderer_amd", "csrc", "fr_export8_table.inc")).read()
    # this host's libm: at most a float apart, at a handful of thresholds
    h = fr.export8_thresholds(host_powf=True)
    d = h[:256].view(np.uint32).astype(np.int64) - t[:256].view(np.uint32).astype(np.int64)
    print("host powf differs from the correctly rounded power at thresholds", np.nonzero(d)[0].tolist(), "by", d[d != 0].tolist(), "ulp")
    assert np.abs(d).max() <= 1 and (d != 0).sum() <= 8
    # sanity of the table's values in double: the threshold's byte is b, its predecessor's is below b
    below = (t[1:256].view(np.uint32) - 1).view(np.float32)
    assert (np.float32(255.0) * np.power(t[1:256].astype(np.float64), 1 / 2.2) >= np.arange(1, 256) - 1e-3).all()
    assert (np.float32(255.0) * np.power(below.astype(np.float64), 1 / 2.2) < np.arange(1, 256) + 1e-3).all()


def test_hot_kernels_keep_their_register_budget(fr):
    """The launcher sizes its persistent grids for a resident set (tile pass 5, lane pool 6 workgroups of 4 waves per
    CU): a change that pushes a hot kernel over its VGPR budget silently leaves workgroups non-resident (an
    experimental option once cost the lane pool 12 VGPRs and 34 SGPR spills: -5 % to -13 % on every workload).
    Recompiles the device code with resource remarks and checks the kernels the defaults launch."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "fractalrenderer_amd", "csrc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                          "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-c", os.path.join(csrc, "fr_device.hip"),
                          "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    usage = {}
    cur = None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = usage.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    budget = {   # mangled name: (max VGPRs, min waves/SIMD, max SGPR spills)
        # lane pool: queue parameters, stream description and output planes are re-read from the kernel arguments where
        # they are used (kargs()), so nothing spills -- v_readlane / v_writelane are VALU issue slots
        # (VGPRs come in granules of 16 on gfx950; round 4's deferred-escape ring -- 22 KB of LDS per workgroup -- and its replay
        # registers put the fp64 pool at 70 VGPRs = 6 resident workgroups per CU; 5, 6 and 8 waves per SIMD measured the same)
        "_ZN2fr11pool_kernelIdLi0ELb0EEEvNS_10LaunchArgsE": (80, 6, 0),    # fp64 Mandelbrot lane pool (C2/C4/C5)
        "_ZN2fr11pool_kernelIdLi0ELb1EEEvNS_10LaunchArgsE": (80, 6, 12),   # ... with cycle closing (the default; the ring's
                                                                            # counters pushed 12 SGPRs out: measured -6 % all the same)
        "_ZN2fr11pool_kernelIfLi1ELb0EEEvNS_10LaunchArgsE": (64, 6, 0),    # fp32 Julia lane pool (C3)
        "_ZN2fr11pool_kernelIfLi1ELb1EEEvNS_10LaunchArgsE": (64, 6, 4),    # ... with cycle closing (the default)
        # lean tile kernel, two sub-tiles per trip (the default tile pass): 5 workgroups per CU = 5 waves per SIMD
        # (with the prologue in it -- round 4 -- the register allocator settles the fp64 kernel at 80 VGPRs / 17 spilled SGPRs
        # instead of 92 / 11: its tile pass measured 10 % shorter on C2, 7 % on C5)
        "_ZN2fr16tile_lean_kernelIdLi0ELb0ELi2EEEvNS_10LaunchArgsE": (96, 5, 20),  # fp64 Mandelbrot, staged (C2/C4/C5)
        "_ZN2fr16tile_lean_kernelIdLi0ELb1ELi2EEEvNS_10LaunchArgsE": (96, 5, 20),  # one-pass frames (C1), cycle closing
        "_ZN2fr16tile_lean_kernelIfLi1ELb0ELi2EEEvNS_10LaunchArgsE": (64, 6, 0),   # fp32 Julia (C3): 6 workgroups per CU
        # one-pass fp32 Mandelbrot with cycle closing: 1080p at max_iter 256, the reference's interactive default (75 VGPRs = 6
        # waves per SIMD until round 4's prologue changed what the allocator does with the Mandelbrot instantiations: 51 = 7,
        # -14 % on that frame)
        "_ZN2fr16tile_lean_kernelIfLi0ELb1ELi2EEEvNS_10LaunchArgsE": (64, 7, 0),
        "_ZN2fr16tile_lean_kernelIdLi0ELb0ELi1EEEvNS_10LaunchArgsE": (64, 5, 16),   # one sub-tile per trip
        # stripe shading through the lean kernels (kernel code 3: the z of the last update rides along): 5 resident workgroups
        "_ZN2fr16tile_lean_kernelIdLi3ELb0ELi2EEEvNS_10LaunchArgsE": (96, 5, 20),
        "_ZN2fr11pool_kernelIdLi3ELb0EEEvNS_10LaunchArgsE": (96, 5, 4),
        # general tile kernel (SSAA, other sub-tile shapes, strips that are not whole sub-tile rows)
        "_ZN2fr11tile_kernelIdLi0ELi3ELb0ELb0ELb1EEEvNS_10LaunchArgsE": (96, 5, 48),
        "_ZN2fr11tile_kernelIdLi0ELi3ELb0ELb0ELb0EEEvNS_10LaunchArgsE": (96, 5, 48),
        "_ZN2fr11tile_kernelIfLi1ELi3ELb0ELb0ELb0EEEvNS_10LaunchArgsE": (64, 5, 24),
        # the effects variant (orbit trap / stripes): its fp64 atan2 + sin epilogue is CALLED since round 4 (inlined it held the
        # kernel at 156 VGPRs = 3 waves per SIMD)
        "_ZN2fr11tile_kernelIdLi0ELi3ELb1ELb0ELb0EEEvNS_10LaunchArgsE": (112, 5, 24),
    }
    for name, (max_vgpr, min_occ, max_spill) in budget.items():
        u = usage.get(name)
        assert u, "kernel %s not found in the resource remarks" % name
        assert u["ScratchSize [bytes/lane]"] == 0, (name, u)
        assert u["VGPRs"] <= max_vgpr and u["Occupancy [waves/SIMD]"] >= min_occ and u["SGPRs Spill"] <= max_spill, (name, u)


def test_host_code_under_sanitizers(golden, tmp_path, monkeypatch):
    """The C host side (fr_host.c, fr_franim.c, fr_frameio.c, fr_zoompath.c) built with -fsanitize=address,undefined and with
    -fsanitize=thread (CPU builds only: GPU sanitizers are not available on the pool) and driven by
    tests/c_client/fuzz_host.c: truncated and corrupted .franim inputs, interpolation outside the keyframe range,
    save / re-parse, multi-band PNG writes on 8 worker threads."""
    import subprocess
    csrc = os.path.join(ROOT, "fractalrenderer_amd", "csrc")
    monkeypatch.setenv("FR_PNG_THREADS", "8")
    for tag, flags in (("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
                       ("tsan", ["-fsanitize=thread"])):
        exe = str(tmp_path / ("fuzz_host_" + tag))
        cmd = ["gcc", "-std=c11", "-g", "-O1", *flags, "-fno-omit-frame-pointer",
               "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
               os.path.join(csrc, "fr_host.c"), os.path.join(csrc, "fr_franim.c"), os.path.join(csrc, "fr_frameio.c"),
               os.path.join(csrc, "fr_zoompath.c"),
               os.path.join(ROOT, "tests", "c_client", "fuzz_host.c"), "-o", exe, "-lm", "-lz", "-lpthread"]
        out = subprocess.run(cmd, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        run = subprocess.run([exe, golden["franim"], str(tmp_path)], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0 and "ERROR" not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr \
            and "runtime error" not in run.stderr, (tag, run.stderr[-3000:])
        assert run.stdout.startswith("parsed ok")


def test_bench_workloads_and_cpu_baseline_leg(oracle):
    """bench.py's workload table is well formed (the tools import it too) and its cpu_baseline leg -- the oracle timed
    on the host cores -- returns the contract's object on a bounded sample (here: the C1 frame)."""
    import bench
    for name, w in bench.WORKLOADS.items():
        assert {"desc", "fractal", "precision", "W", "H", "state"} <= set(w), name
        assert w["fractal"] in ("Mandelbrot", "JuliaSet", "Deep_Zoom") and w["precision"] in ("F64", "F32"), name
        if "kernel" in w:           # colorize / export entries: one launch of that kernel per step, an HBM roofline
            assert w["kernel"] in ("colorize", "export8", "export16") and w["bytes_per_pixel"] in (24, 19, 22), name
        assert w["state"]["max_iterations"] >= 1 and w["W"] > 0 and w["H"] > 0
    assert bench.WORKLOADS["c2"]["W"] == bench.WORKLOADS["c2"]["H"] == 4096
    assert bench.WORKLOADS["c2"]["state"] == {"max_iterations": 1024} and bench.WORKLOADS["c2"]["precision"] == "F64"
    w = dict(bench.WORKLOADS["c1"], cpu_passes=1)
    b = bench.cpu_baseline(w)
    assert b["kind"] == "port" and b["unit"] == "Mpixels/s" and b["cores"] >= 1 and b["value"] > 0
    assert "sample" in b and "oracle" in b["sample"]
    assert b["single_thread"]["cores"] == 1 and b["single_thread"]["value"] > 0
    # the Deep_Zoom workload goes through the same leg (fractal 5 of the oracle)
    wd = dict(bench.WORKLOADS["deepzoom"], W=64, H=48, cpu_rows=48, cpu_passes=1)
    wd["state"] = dict(wd["state"], max_iterations=200)
    assert bench.cpu_baseline(wd)["value"] > 0


def test_colorize_supported_stops_where_a_float_nu_can_round_up_to_max_iter(fr):
    """fr_colorize_supported tells the multi-GPU exchange that it may ship the nu plane and recolour it, reading
    'interior' off nu == max_iter.  In fp32 a sample escaping at i = max_iter - 1 has nu = RN(max_iter - mu): once the
    float spacing at max_iter reaches 1 (max_iter >= 2^23) that IS max_iter and the pixel would be painted as interior.
    The library must refuse the shortcut from 2^22 on (one binade of margin) in fp32 and keep it in fp64."""
    import numpy as np
    L = fr.lib()
    for prec, mi, want in ((fr.Precision.F32, 1 << 22, 1), (fr.Precision.F32, (1 << 22) + 1, 0), (fr.Precision.F32, 1 << 24, 0),
                           (fr.Precision.F64, 1 << 24, 1), (fr.Precision.F32, 2048, 1)):
        for ft in (fr.FractalType.Mandelbrot, fr.FractalType.JuliaSet, fr.FractalType.BurningShip):
            p = fr.FractalState(max_iterations=mi).to_params(ft, prec)
            assert L.fr_colorize_supported(C.byref(p)) == want, (prec, mi, ft)
    # the arithmetic behind the limit: smallest mu the supported bailouts allow is ~0.32 (Mandelbrot, bailout 2.5:
    # mu = log2(log2 |z|) with |z| just above 2.5)
    mu = np.log2(np.log2(2.5))
    assert 0.3 < mu < 0.5
    assert np.float32(np.float32(1 << 22) - np.float32(mu)) < np.float32(1 << 22)        # still below: representable
    assert np.float32(np.float32(1 << 24) - np.float32(mu)) == np.float32(1 << 24)       # rounds up: the hazard


# ---- deep-zoom zoom paths (src/deep_zoom_system.cpp:454-556) ---------------------------------------------------
def _zoom_path_model(path, state, steps):
    """Pure-Python restatement of DeepZoomManager::update_animation + interpolate_to_keyframe (:487-556), float32 time
    arithmetic as the reference's floats, double centre / zoom: (cx, cy, zoom, animating, progress, orbit_dirty) per step."""
    import math
    f32 = np.float32
    cur, t = 0, f32(0.0)
    animating, progress = len(path) > 0, f32(0.0)
    cx, cy, z = state
    out = []
    for dt in steps:
        dirty = False
        if not path or cur >= len(path):
            animating = False
        else:
            t = f32(t + f32(dt))
            kx, ky, kz, kd = path[cur]
            if t >= f32(kd):
                cx, cy, z = kx, ky, kz
                cur += 1
                t = f32(0.0)
                dirty = True
                if cur >= len(path):
                    animating, progress = False, f32(1.0)
            else:
                tt = float(f32(t / f32(kd)))
                if 0 < cur < len(path):
                    px, py, pz, _ = path[cur - 1]
                    lz = math.log(pz) + tt * (math.log(kz) - math.log(pz))
                    cx, cy, z = px + tt * (kx - px), py + tt * (ky - py), math.exp(lz)
                total = f32(0.0); elapsed = f32(0.0)
                for i, k in enumerate(path):
                    total = f32(total + f32(k[3]))
                    if i < cur:
                        elapsed = f32(elapsed + f32(k[3]))
                elapsed = f32(elapsed + t)
                progress = f32(elapsed / total) if total > 0 else f32(1.0)
        out.append((cx, cy, z, animating, float(progress), dirty))
    return out


def test_zoom_path_follows_the_reference_state_machine(fr):
    """zoomTo + update_animation step by step against the restated state machine: the start keyframe (duration 0) is
    reached at the first update, then centre linear / zoom in log space, the target hit exactly when its duration has
    passed, zoom_animating / zoom_progress as DeepZoomState reports them."""
    st = fr.FractalState()                                   # centre (-0.5, 0), zoom 3
    kf = fr.DeepZoomPath.preset("Seahorse")
    assert (kf.center_x, kf.center_y, kf.zoom, kf.duration) == (-0.743643887037151, 0.13182590420533, 1e-6, 5.0)
    assert fr.DeepZoomPath.preset(1).zoom == 1e-8 and fr.DeepZoomPath.preset(2).duration == 10.0
    z = fr.DeepZoomPath(st)
    z.zoom_to(kf.center_x, kf.center_y, kf.zoom, kf.duration)
    assert z.zoom_animating and z.zoom_progress == 0.0
    steps = [0.016, 1.0, 0.5, 1.25, 2.0, 0.2, 0.1, 0.3]
    model = _zoom_path_model([(-0.5, 0.0, 3.0, 0.0), (kf.center_x, kf.center_y, kf.zoom, kf.duration)], (-0.5, 0.0, 3.0), steps)
    for dt, (cx, cy, zoom, anim, prog, dirty) in zip(steps, model):
        got_dirty = z.update_animation(dt)
        assert (st.center_x, st.center_y) == (cx, cy) and st.zoom == zoom, (dt, st, (cx, cy, zoom))
        assert z.zoom_animating == anim and abs(z.zoom_progress - prog) < 1e-7 and got_dirty == dirty
    assert not z.zoom_animating and z.zoom_progress == 1.0
    assert (st.center_x, st.center_y, st.zoom) == (kf.center_x, kf.center_y, kf.zoom)      # landed on the target exactly
    # log-space: half way in time is the geometric mean of the zooms
    z2 = fr.DeepZoomPath(fr.FractalState(zoom=4.0))
    z2.play_zoom_path([fr.ZoomKeyframe(0.0, 0.0, 4.0, 0.0), fr.ZoomKeyframe(1.0, -1.0, 1e-4, 2.0), fr.ZoomKeyframe(1.0, -1.0, 1e-2, 1.0)])
    z2.update_animation(0.0)
    z2.update_animation(1.0)
    assert abs(z2.state.zoom - 0.02) < 1e-15 and z2.state.center_x == 0.5 and abs(z2.zoom_progress - 1.0 / 3.0) < 1e-7
    z2.update_animation(1.0)                                  # reaches keyframe 1
    z2.update_animation(0.5)                                  # half way to keyframe 2: zoom back out, log space
    assert abs(z2.state.zoom - 1e-3) < 1e-18 and abs(z2.zoom_progress - 2.5 / 3.0) < 1e-7
    # an empty path plays nothing; a non-positive zoom is refused (log space)
    z3 = fr.DeepZoomPath()
    z3.play_zoom_path([])
    z3.update_animation(1.0)
    assert not z3.zoom_animating and z3.state.zoom == 3.0
    with pytest.raises(fr.FractalRendererError):
        z3.play_zoom_path([fr.ZoomKeyframe(0, 0, 0.0, 1.0)])
