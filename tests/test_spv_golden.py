"""The oracle against vectors produced by EXECUTING the reference's compiled shaders.

tests/golden/spv_frames.npz holds, per case of tests/spv_cases.py, what shaders/<shader>.comp.spv of the
reference writes to its storage image when it is run invocation by invocation through the SPIR-V
interpreter of tests/golden/spirv_interp.py (generator: tests/golden/make_spv_golden.py; the binaries are
read from the reference at generation time and are not part of this repository).  This pins the CPU
restatement in oracle/ to the reference's own artefact:

  * escape index: bit-exact (the iteration only uses + - * and comparisons, one rounding per operation on
    both sides);
  * smooth value on escaped samples: |d| <= 4 ulp32 + 4e-6 (glibc logf vs numpy's float32 log);
  * written texel: |d| <= 5e-6 per channel (logf/powf/sinf/expf implementations; observed <= 8e-7).
"""
import json
import os
import struct

import numpy as np
import pytest

from spv_cases import SPV_CASES

HERE = os.path.dirname(os.path.abspath(__file__))
RGBA_TOL = 5e-6


@pytest.fixture(scope="module")
def spv():
    return np.load(os.path.join(HERE, "golden", "spv_frames.npz"))


def test_fixture_is_complete(spv):
    meta = json.loads(str(spv["__meta__"]))
    assert sorted(meta["sha256"]) == ["burning_ship.comp.spv", "julia.comp.spv", "mandelbrot.comp.spv",
                                      "test_deep_zoom.comp.spv"]
    for name, (shader, p, W, H) in SPV_CASES.items():
        rgba = spv[name + "/rgba"]
        assert rgba.shape == (H, W, 4) and rgba.dtype == np.float32 and np.all(rgba[..., 3] == 1.0), name
        assert np.isfinite(rgba).all() and rgba[..., :3].max() > 0.05, name
        if p.aa == 1:
            it = spv[name + "/iter"]
            assert it.shape == (H, W) and it.min() >= 0 and it.max() <= p.max_iterations
            assert (it < p.max_iterations).any(), name             # every case has escaping pixels
    # the set covers every palette of the three escape-time shaders, every interior style and both effects
    modes = {(s, p.palette_mode) for s, p, _, _ in SPV_CASES.values()}
    assert {("mandelbrot", m) for m in range(6)} <= modes
    assert {(s, m) for s in ("julia", "burning_ship") for m in range(10)} <= modes
    assert {(s, p.interior_style) for s, p, _, _ in SPV_CASES.values()} >= {(s, k) for s in ("mandelbrot", "burning_ship")
                                                                            for k in range(4)}


@pytest.mark.parametrize("name", sorted(SPV_CASES))
def test_oracle_matches_the_executed_reference_shader(oracle, spv, name):
    shader, p, W, H = SPV_CASES[name]
    f = oracle.render(p, W, H)
    want = spv[name + "/rgba"]
    d = np.abs(f.rgba - want)
    assert d.max() <= RGBA_TOL, "texel differs by %g at %s" % (d.max(), np.unravel_index(d.argmax(), d.shape))
    if p.aa != 1:
        return
    it, sm = spv[name + "/iter"], spv[name + "/smooth"]
    assert np.array_equal(f.iter, it), "%d escape indices differ" % int((f.iter != it).sum())
    esc = (it < p.max_iterations) & ~np.isnan(sm)
    assert esc.sum() == (it < p.max_iterations).sum()              # a smooth value exists for every escaped sample
    ulp = np.spacing(np.maximum(np.abs(sm[esc]), 1.0).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(f.nu[esc] - sm[esc].astype(np.float64)) <= 4 * ulp + 4e-6)


def test_fp64_restatement_is_the_same_algorithm_at_wider_precision(oracle, spv):
    """The fp64 planes (the north-star arithmetic) cannot be bit-compared with an fp32 shader: orbits near the
    boundary are chaotic and legitimately part ways.  What must hold is that they are the same function away from
    it: nearly all escape indices coincide and, where they do, colours differ by float round-off amplified by the
    orbit, not by a different formula (median difference at the 1e-6 level)."""
    import dataclasses
    agree, total, medians = 0, 0, []
    for name, (shader, p, W, H) in SPV_CASES.items():
        if shader == "test_deep_zoom" or p.aa != 1:         # Deep_Zoom has no fp64 variant in the reference
            continue
        if name.endswith("_long_orbits"):                   # chaos on purpose: fp32 and fp64 part ways there by design
            continue
        f = oracle.render(dataclasses.replace(p, precision=1), W, H)
        same = f.iter == spv[name + "/iter"]
        assert same.mean() >= 0.85, name
        d = np.abs(f.rgba - spv[name + "/rgba"]).max(axis=2)[same]
        medians.append(float(np.median(d)))
        agree += int(same.sum())
        total += same.size
    assert agree / total >= 0.98
    assert max(medians) <= 1e-4 and float(np.median(medians)) <= 2e-6


# ---- the interpreter itself ----------------------------------------------------------------------------------
def _assemble(instructions, bound):
    words = [0x07230203, 0x00010000, 0, bound, 0]
    for op, *operands in instructions:
        flat = []
        for o in operands:
            if isinstance(o, str):
                b = o.encode() + b"\0"
                b += b"\0" * (-len(b) % 4)
                flat += list(struct.unpack("<%dI" % (len(b) // 4), b))
            elif isinstance(o, float):
                flat.append(struct.unpack("<I", struct.pack("<f", o))[0])
            else:
                flat.append(o & 0xFFFFFFFF)
        words.append(((len(flat) + 1) << 16) | op)
        words += flat
    return struct.pack("<%dI" % len(words), *words)


def test_interpreter_on_a_hand_assembled_module(tmp_path):
    """for (i = 0; i < gid.x; ++i) acc = fma(acc, 0.5, 1.0);  imageStore(img, gid.xy, vec4(acc, sqrt(acc), i, 1))"""
    from golden.spirv_interp import Cell, Invocation, Module
    (void, fn, f32, i32, u32, v3u, v2i, v4f, boolt, img, p_img, p_in, p_fi, p_ff,
     c0, c1, ch, c1f, c0f, glsl, main, gid, image, vi, vacc, l0, lh, lb, lc, lm) = range(1, 31)
    t = list(range(31, 60))
    ins = [
        (17, 1), (11, glsl, "GLSL.std.450"), (14, 0, 1), (15, 5, main, "main", gid),
        (5, gid, "gl_GlobalInvocationID"), (5, image, "image"), (5, vi, "i"), (5, vacc, "acc"), (5, main, "main"),
        (19, void), (33, fn, void), (22, f32, 32), (21, i32, 32, 1), (21, u32, 32, 0), (23, v3u, u32, 3),
        (23, v2i, i32, 2), (23, v4f, f32, 4), (20, boolt), (25, img, f32, 1, 0, 0, 0, 2, 1),
        (32, p_img, 0, img), (32, p_in, 1, v3u), (32, p_fi, 7, i32), (32, p_ff, 7, f32),
        (43, i32, c0, 0), (43, i32, c1, 1), (43, f32, ch, 0.5), (43, f32, c1f, 1.0), (43, f32, c0f, 0.0),
        (59, p_img, image, 0), (59, p_in, gid, 1),
        (54, void, main, 0, fn), (248, l0),
        (59, p_fi, vi, 7), (59, p_ff, vacc, 7), (62, vi, c0), (62, vacc, c0f), (249, lh),
        (248, lh), (246, lm, lc, 0), (61, i32, t[0], vi), (61, v3u, t[1], gid), (81, u32, t[2], t[1], 0),
        (124, i32, t[3], t[2]), (177, boolt, t[4], t[0], t[3]), (250, t[4], lb, lm),
        (248, lb), (61, f32, t[5], vacc), (12, f32, t[6], glsl, 50, t[5], ch, c1f), (62, vacc, t[6]), (249, lc),
        (248, lc), (61, i32, t[7], vi), (128, i32, t[8], t[7], c1), (62, vi, t[8]), (249, lh),
        (248, lm), (61, f32, t[9], vacc), (12, f32, t[10], glsl, 31, t[9]), (61, i32, t[11], vi),
        (111, f32, t[12], t[11]), (80, v4f, t[13], t[9], t[10], t[12], c1f),
        (61, v3u, t[14], gid), (79, v3u, t[15], t[14], t[14], 0, 1, 2), (81, u32, t[16], t[15], 0), (81, u32, t[17], t[15], 1),
        (124, i32, t[18], t[16]), (124, i32, t[19], t[17]), (80, v2i, t[20], t[18], t[19]),
        (61, img, t[21], image), (99, t[21], t[20], t[13]), (253,), (56,),
    ]
    path = tmp_path / "loop.spv"
    path.write_bytes(_assemble(ins, 64))
    m = Module(str(path))
    assert m.entry == main and m.global_named("image") == image
    for n in (0, 1, 5):
        inv = Invocation(m, {gid: Cell([n, 7, 0]), image: Cell(None)}, (8, 8), probe=("main:acc",)).run()
        acc = np.float32(0.0)
        for _ in range(n):
            acc = np.float32(acc * np.float32(0.5) + np.float32(1.0))
        (_, coord, texel), = inv.stores
        assert coord == [n, 7] and texel == [acc, np.float32(np.sqrt(acc)), np.float32(n), np.float32(1.0)]
        assert inv.probes["main:acc"] == acc


def test_interpreter_float_semantics():
    from golden import spirv_interp as S
    f = np.float32
    # one rounding for fma: (1 + 2^-23)(1 - 2^-23) - 1 = -2^-46 exactly; a*b alone rounds to 1
    a, b = f(1) + f(2.0 ** -23), f(1) - f(2.0 ** -23)
    assert a * b == f(1) and S._fma32(a, b, f(-1)) == f(-2.0 ** -46)
    assert S._fma32(f(3), f(4), f(5)) == f(17) and S._fma32(f(0), f(4), f(0)) == f(0)
    g = S._GLSL
    assert g[10](f(-0.25)) == f(0.75) and g[10]([f(1.5), f(2.0)]) == [f(0.5), f(0.0)]        # Fract
    assert g[43](f(3), f(0), f(1)) == f(1) and g[46](f(2), f(4), f(0.25)) == f(2.5)         # FClamp, FMix
    assert g[49](f(0), f(2), f(1)) == f(0.5) and g[49](f(0), f(2), f(9)) == f(1)            # SmoothStep
    assert g[37](f(1), f(2)) == f(1) and g[40](f(1), f(2)) == f(2) and g[42](3, -4) == 3     # FMin, FMax, SMax
    assert S._length([f(3), f(4)]) == f(5)
    assert S._i32(0xFFFFFFFF) == -1 and S._i32(1 << 31) == -(1 << 31)
    assert isinstance(g[28](f(2)), np.float32) and g[28](f(2)) == np.log(f(2))              # float32 log, not double
