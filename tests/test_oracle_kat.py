"""Pins the CPU oracle: analytic known answers, an independent numpy restatement, a
high-precision (mpmath) check of nu, and the committed golden fixtures.

The reference holds no tests or golden vectors for this path, so
these known answers are the build's own; the vectors that come from the reference's own
artefact (its compiled shaders, executed) are checked in tests/test_spv_golden.py.  CPU only.
"""
import math

import numpy as np
import pytest

from cases import CASES, MANDEL_PALETTES, JULIA_PALETTES
from oracle import np_restatement as NP


def centre_pixel(O, **kw):
    """2x2 frame: pixel (W/2, H/2) = (1,1) maps exactly to the centre (shaders/mandelbrot.comp:150)."""
    f = O.render(O.OracleParams(**kw), 2, 2, threads=1)
    return int(f.iter[1, 1]), float(f.nu[1, 1])


# ---- (i) analytic known answers, bailout 4 -> test |z|^2 > 16 ------------------------------
@pytest.mark.parametrize("prec", [0, 1])
def test_kat_interior_points(oracle, prec):
    for c in [(0.0, 0.0), (-2.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (-0.5, 0.0)]:
        it, nu = centre_pixel(oracle, center_x=c[0], center_y=c[1], max_iterations=200, precision=prec)
        assert it == 200 and nu == 200.0, c      # c=-2: orbit 0,-2,2,2..: |z|^2 = 4 is not > 16


@pytest.mark.parametrize("prec,tol", [(0, 2e-6), (1, 1e-14)])
def test_kat_escaping_points(oracle, prec, tol):
    # c = 1: z = 1, 2, 5 -> escapes at the update with index i = 2, |z| = 5
    it, nu = centre_pixel(oracle, center_x=1.0, center_y=0.0, precision=prec)
    assert it == 2 and abs(nu - (3 - math.log2(math.log2(5)))) < tol
    # c = 2: z = 2, 6 -> i = 1
    it, nu = centre_pixel(oracle, center_x=2.0, center_y=0.0, precision=prec)
    assert it == 1 and abs(nu - (2 - math.log2(math.log2(6)))) < tol
    # c = 100: z = 100 -> i = 0
    it, nu = centre_pixel(oracle, center_x=100.0, center_y=0.0, precision=prec)
    assert it == 0 and abs(nu - (1 - math.log2(math.log2(100)))) < tol


@pytest.mark.parametrize("prec,tol", [(0, 2e-6), (1, 1e-14)])
def test_kat_julia(oracle, prec, tol):
    # Julia c = 0, z0 = (3, 0): z1 = 9, |z|^2 = 81 > 16 at i = 0; nu uses log(bailout) (shaders/julia.comp:238)
    # 2x2 frame, aspect 1: pixel (1,1) -> uv = 0.5 -> z0 = centre
    it, nu = centre_pixel(oracle, fractal=1, center_x=3.0, center_y=0.0, julia_c_real=0.0, julia_c_imag=0.0, precision=prec)
    expect = 0 + 1 - math.log(math.log(81.0) / math.log(4.0)) / math.log(2.0)
    assert it == 0 and abs(nu - expect) < tol
    # z0 = 0, c = 0 -> fixed point, interior, black
    f = oracle.render(oracle.OracleParams(fractal=1, center_x=0.0, julia_c_real=0.0, julia_c_imag=0.0, precision=prec), 2, 2)
    assert f.iter[1, 1] == 256 and np.all(f.rgba[1, 1, :3] == 0.0) and f.rgba[1, 1, 3] == 1.0


def test_nu_equals_mandelbrot_form_only_for_bailout_4(oracle):
    """a2 note: Julia's nu = iter+1-log(log|z|^2/log B)/log2 equals a1's form only when B = 4."""
    kw = dict(fractal=1, julia_c_real=0.0, julia_c_imag=0.0, center_x=3.0, center_y=0.5, max_iterations=50)
    for bailout, same in ((4.0, True), (8.0, False)):
        j = oracle.render(oracle.OracleParams(bailout=bailout, **kw), 2, 2)
        r2 = j.zre[1, 1] ** 2 + j.zim[1, 1] ** 2
        a1 = j.iter[1, 1] + 1 - math.log(math.log(r2) / 2 / math.log(2)) / math.log(2)    # shaders/mandelbrot.comp:174-176
        assert (abs(a1 - j.nu[1, 1]) < 1e-13) == same


# ---- (ii) viewport known answers -------------------------------------------------------------
def test_viewport_mandelbrot(oracle):
    W, H = 64, 32
    p = oracle.OracleParams(center_x=-0.75, center_y=0.25, zoom=2.0, max_iterations=1)
    f = oracle.render(p, W, H)          # one update from z = 0: final z == c for every pixel
    assert f.zre[H // 2, W // 2] == -0.75 and f.zim[H // 2, W // 2] == 0.25
    # pixel (0,0) -> centre - (0.5*W/H, 0.5)*zoom ; y grows downward, no pixel-centre offset
    assert f.zre[0, 0] == -0.75 - 0.5 * W / H * 2.0 and f.zim[0, 0] == 0.25 - 0.5 * 2.0
    assert f.zre[0, 1] - f.zre[0, 0] == pytest.approx(2.0 / H, rel=1e-15)


def test_viewport_julia(oracle):
    W, H = 64, 32
    p = oracle.OracleParams(fractal=1, center_x=0.5, center_y=-0.25, zoom=2.0, max_iterations=1,
                            julia_c_real=0.0, julia_c_imag=0.0)
    f = oracle.render(p, W, H)          # z1 = z0^2 with c = 0
    z0x, z0y = 0.5 + (0 / W - 0.5) * 2.0 * (W / H), -0.25 + (0 / H - 0.5) * 2.0      # shaders/julia.comp:222-225
    assert f.zre[0, 0] == z0x * z0x - z0y * z0y and f.zim[0, 0] == 2.0 * z0x * z0y
    assert f.zre[H // 2, W // 2] == 0.5 * 0.5 - 0.25 * 0.25


# ---- (iii) independent numpy restatement -------------------------------------------------------
@pytest.mark.parametrize("name", ["c1_mandel_f64_default", "c2_mandel_f64_mi1024_ragged", "c2_mandel_f32_mi1024",
                                  "seahorse_0008_f64", "mandel_small_bailout", "mandel_far_exterior"])
def test_numpy_restatement_mandelbrot(oracle, name):
    p, W, H = CASES[name]
    T = np.float64 if p.precision == 1 else np.float32
    it, nu, zx, zy = NP.mandelbrot(W, H, p.center_x, p.center_y, p.zoom, p.max_iterations, p.bailout, T)
    f = oracle.render(p, W, H)
    assert np.array_equal(it, f.iter)
    assert np.array_equal(zx.astype(np.float64), f.zre) and np.array_equal(zy.astype(np.float64), f.zim)
    tol = 1e-11 if p.precision == 1 else 4e-4       # log implementations differ (numpy SIMD vs libm)
    assert np.abs(nu.astype(np.float64) - f.nu).max() <= tol


@pytest.mark.parametrize("name", ["ship_f64_overview", "ship_f32_the_ship", "ship_f64_ragged_mi2048",
                                  "ship_small_bailout_f64"])
def test_numpy_restatement_burning_ship(oracle, name):
    p, W, H = CASES[name]
    T = np.float64 if p.precision == 1 else np.float32
    it, nu, zx, zy = NP.burning_ship(W, H, p.center_x, p.center_y, p.zoom, p.max_iterations, p.bailout, T)
    f = oracle.render(p, W, H)
    assert np.array_equal(it, f.iter)
    assert np.array_equal(zx.astype(np.float64), f.zre) and np.array_equal(zy.astype(np.float64), f.zim)
    tol = 1e-11 if p.precision == 1 else 4e-4
    assert np.abs(nu.astype(np.float64) - f.nu).max() <= tol


def test_burning_ship_kat(oracle):
    """Hand-checkable orbits of z <- (|Re z| + i|Im z|)^2 + c (shaders/burning_ship.comp:241-245):
    c = -1 -> 0, -1, 0, -1 ... (bounded);  c = -1.75 lies on the real antenna (bounded, same as the
    Mandelbrot real axis since x^2 ignores the fold);  c = 1+i -> z1 = 1+i, z2 = 1+3i, z3 = -7+7i:
    |z3|^2 = 98 > bailout^2 = 16 at loop index 2."""
    for cx, cy, expect in ((-1.0, 0.0, 64), (-1.75, 0.0, 64), (1.0, 1.0, 2), (0.0, -1.0, 64)):
        p = oracle.OracleParams(fractal=2, center_x=cx, center_y=cy, zoom=1e-9, max_iterations=64)
        f = oracle.render(p, 2, 2)
        assert int(f.iter[1, 1]) == expect, (cx, cy, f.iter)
    # escape index 2 at c = 1 + i: nu = 2 + 1 - log2(log(98)/log(4))
    f = oracle.render(oracle.OracleParams(fractal=2, center_x=1.0, center_y=1.0, zoom=1e-12, max_iterations=64), 2, 2)
    assert abs(f.nu[1, 1] - (3.0 - np.log2(np.log(98.0) / np.log(4.0)))) < 1e-9
    # folding breaks the conjugate symmetry Mandelbrot has: c and conj(c) give different orbits
    up = oracle.render(oracle.OracleParams(fractal=2, center_x=-0.5, center_y=-0.6, zoom=1e-9, max_iterations=256), 2, 2)
    dn = oracle.render(oracle.OracleParams(fractal=2, center_x=-0.5, center_y=0.6, zoom=1e-9, max_iterations=256), 2, 2)
    assert up.iter[1, 1] != dn.iter[1, 1]


@pytest.mark.parametrize("name", ["c3_julia_f32_centre0", "c3_julia_f32_default_centre", "julia_f64_default_c",
                                  "julia_c_outside_bailout"])
def test_numpy_restatement_julia(oracle, name):
    p, W, H = CASES[name]
    T = np.float64 if p.precision == 1 else np.float32
    it, nu, zx, zy = NP.julia(W, H, p.center_x, p.center_y, p.zoom, p.max_iterations,
                              p.julia_c_real, p.julia_c_imag, p.bailout, T)
    f = oracle.render(p, W, H)
    assert np.array_equal(it, f.iter)
    assert np.array_equal(zx.astype(np.float64), f.zre) and np.array_equal(zy.astype(np.float64), f.zim)
    tol = 1e-11 if p.precision == 1 else 4e-4
    assert np.abs(nu.astype(np.float64) - f.nu).max() <= tol


def test_nu_high_precision(oracle):
    """nu from the oracle's own z-at-escape recomputed with 40 digits; and, for short orbits,
    the whole orbit re-iterated in 60-digit arithmetic from the exact double c."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    p, W, H = CASES["c1_mandel_f64_default"]
    f = oracle.render(p, W, H)
    ys, xs = np.nonzero(f.iter < p.max_iterations)
    rng = np.random.default_rng(7)
    pick = rng.choice(len(ys), 40, replace=False)
    for k in pick:
        y, x = int(ys[k]), int(xs[k])
        r2 = mp.mpf(float(f.zre[y, x])) ** 2 + mp.mpf(float(f.zim[y, x])) ** 2
        nu = f.iter[y, x] + 1 - mp.log(mp.log(r2) / 2 / mp.log(2)) / mp.log(2)
        assert abs(float(nu) - f.nu[y, x]) < 1e-13
    # full re-iteration where the orbit is short enough for fp64 error growth to stay below 1e-9
    short = [(int(y), int(x)) for y, x in zip(ys, xs) if f.iter[y, x] <= 12][:25]
    for y, x in short:
        uvx = (mp.mpf(x) - mp.mpf(W) / 2) / H
        uvy = (mp.mpf(y) - mp.mpf(H) / 2) / H
        c = mp.mpc(mp.mpf(p.center_x) + uvx * p.zoom, mp.mpf(p.center_y) + uvy * p.zoom)
        z = mp.mpc(0)
        for i in range(p.max_iterations):
            z = z * z + c
            if z.real ** 2 + z.imag ** 2 > 16:
                break
        assert i == f.iter[y, x]
        nu = i + 1 - mp.log(mp.log(z.real ** 2 + z.imag ** 2) / 2 / mp.log(2)) / mp.log(2)
        assert abs(float(nu) - f.nu[y, x]) < 1e-9


def test_executed_iteration_count(oracle):
    p, W, H = CASES["c1_mandel_f64_default"]
    f = oracle.render(p, W, H)
    expect = int(np.where(f.iter < p.max_iterations, f.iter.astype(np.int64) + 1, p.max_iterations).sum())
    assert f.executed == expect
    one = oracle.render(p, W, H, threads=1)
    assert one.executed == f.executed and np.array_equal(one.nu, f.nu)      # OpenMP split changes nothing
    rows = oracle.render(p, W, H, y0=10, y1=20)
    assert np.array_equal(rows.nu, f.nu[10:20]) and np.array_equal(rows.rgba, f.rgba[10:20])


# ---- (iv) reference orbit (src/deep_zoom_system.cpp:378-424) ------------------------------------
def test_reference_orbit_kat(oracle):
    o = oracle.reference_orbit(-0.5, 0.0, 300)
    assert len(o) == 300 and o[0, 0] == 0.0 and o[1, 0] == -0.5 and o[2, 0] == -0.25
    o = oracle.reference_orbit(1.0, 0.0, 100)          # 0, 1, 2, 5: |5| > 2 found at index 3 -> length 4
    assert o[:, 0].tolist() == [0.0, 1.0, 2.0, 5.0] and np.all(o[:, 1] == 0.0)
    o = oracle.reference_orbit(0.0, 1.0, 50)           # c = i: 0, i, -1+i, -i, -1+i ... bounded
    assert len(o) == 50 and o[3].tolist() == [0.0, -1.0]


# ---- palettes / post chain ------------------------------------------------------------------------
def test_palette_knots(oracle):
    # fire: pow(t,.7) then fifths; t=0 -> c1, t -> 1 wraps through fract to c1 again (shaders/mandelbrot.comp:60-72,130)
    assert oracle.palette(0, 0, 0.0).tolist() == pytest.approx([0.0, 0.0, 0.1])
    assert oracle.palette(0, 0, 1.0).tolist() == pytest.approx([0.0, 0.0, 0.1])
    assert oracle.palette(0, 0, 0.999).tolist() == pytest.approx([1.0, 1.0, 0.95])
    assert oracle.palette(0, 2, 0.3).tolist() == pytest.approx([0.3, 0.3, 0.3])
    # modes 6..9 exist only in the Julia shader; the Mandelbrot shader falls back to fire
    for m in (6, 7, 9, -1, 100):
        assert np.array_equal(oracle.palette(0, m, 0.37), oracle.palette(0, 0, 0.37))
    assert oracle.palette(1, 9, 0.6).tolist() == pytest.approx([0.6, 0.6, 0.6])
    assert oracle.palette(1, 3, 0.9).tolist() == pytest.approx([1.0, 0.95, 0.7])        # sunset, last segment constant
    assert oracle.palette(1, 6, 0.25).tolist() == pytest.approx([0.5, 0.0, 0.5])        # vaporwave knot c2
    assert np.array_equal(oracle.palette(1, 12, 0.2), oracle.palette(1, 0, 0.2))


def test_post_chain_kat(oracle):
    # identity enhance, ACES(1) = 2.54/3.16, gamma 1/2.2  (shaders/mandelbrot.comp:38-54,235)
    v = oracle.post_chain([1.0, 1.0, 1.0])
    assert v.tolist() == pytest.approx([(2.54 / 3.16) ** (1 / 2.2)] * 3, rel=1e-6)
    assert oracle.post_chain([0.0, 0.0, 0.0]).tolist() == [0.0, 0.0, 0.0]
    # saturation 0 -> gray = dot(c, (.299,.587,.114))
    g = 0.2 * 0.299 + 0.4 * 0.587 + 0.6 * 0.114
    a = (g * (2.51 * g + 0.03)) / (g * (2.43 * g + 0.59) + 0.14)
    assert oracle.post_chain([0.2, 0.4, 0.6], saturation=0.0).tolist() == pytest.approx([a ** (1 / 2.2)] * 3, rel=1e-6)
    # Julia floors brightness at 0.1 (shaders/julia.comp:319)
    assert np.allclose(oracle.post_chain([0.5, 0.5, 0.5], brightness=0.0, julia_floors=1),
                       oracle.post_chain([0.5, 0.5, 0.5], brightness=0.1, julia_floors=0))


def test_export_rgb8_kat(oracle):
    img = np.zeros((3, 2, 4), np.float32)
    img[0, :, :3] = 1.0          # top row white
    img[2, 0, 0] = 0.5
    out = oracle.export_rgb8(img)
    assert out.shape == (3, 2, 3)
    white = int(((2.54 / 3.16) ** (1 / 2.2)) * 255.0)
    assert out[2, 0].tolist() == [white] * 3            # vertical flip: top row lands at the bottom (src/vk_engine.cpp:1359)
    a = (0.5 * (2.51 * 0.5 + 0.03)) / (0.5 * (2.43 * 0.5 + 0.59) + 0.14)
    assert out[0, 0, 0] == int((a ** (1 / 2.2)) * 255.0) and out[0, 0, 1] == 0


# ---- Deep_Zoom (shaders/test_deep_zoom.comp) -----------------------------------------------------------
def test_deep_zoom_restatement_properties(oracle):
    """Known structure of the shader's quirks: the view is 4*zoom/H high, the centre pixel has delta 0, and
    with an empty orbit the loop is a plain fp32 iteration started from z = c."""
    W = H = 64
    # centre pixel: offset 0 -> delta 0 -> dz stays 0 -> z_full = reference orbit itself; c = -0.75+0.1i escapes
    p = oracle.OracleParams(fractal=5, precision=0, center_x=-0.75, center_y=0.1, zoom=100.0, max_iterations=300)
    f = oracle.render(p, W, H)
    orb = oracle.reference_orbit(-0.75, 0.1, 300)
    assert len(orb) < 300                                     # the reference point escapes: orbit trimmed
    # z_full at iteration i is z_ref[i] (+0): escape when |orbit[i]|^2 > 16 in float
    of = orb.astype(np.float32)
    first = next(i for i in range(len(of)) if of[i, 0] * of[i, 0] + of[i, 1] * of[i, 1] > np.float32(16.0)) if \
        any(of[i, 0] * of[i, 0] + of[i, 1] * of[i, 1] > np.float32(16.0) for i in range(len(of))) else None
    if first is not None:
        assert f.iter[H // 2, W // 2] == first
    # view height: pixel (W/2, 0) has offset_y = -0.5 -> delta_y = -0.5 * zoom*4/H
    q = oracle.OracleParams(fractal=5, precision=0, center_x=0.0, center_y=0.0, zoom=64.0, max_iterations=1, use_perturbation=0)
    g = oracle.render(q, W, H)      # empty orbit, one update from z = c: z1 = c^2 + c
    cy = np.float32(-0.5) * (np.float32(64.0) * (np.float32(4.0) / np.float32(H)))
    assert g.zre[0, W // 2] == float(np.float32(0.0) * 0 - cy * cy) and g.zim[0, W // 2] == float(cy)
    # interior is black, alpha 1, nu = max_iter
    r = oracle.render(oracle.OracleParams(fractal=5, precision=0, center_x=-0.1, center_y=0.0, zoom=1.0, max_iterations=50), 8, 8)
    assert np.all(r.iter == 50) and np.all(r.rgba[..., :3] == 0) and np.all(r.nu == 50)


# ---- golden fixtures ----------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_frames(oracle, golden, name):
    p, W, H = CASES[name]
    f = oracle.render(p, W, H)
    g = golden["frames"]
    assert np.array_equal(f.iter, g[name + "/iter"])
    assert np.array_equal(f.nu, g[name + "/nu"])
    assert np.array_equal(f.rgba, g[name + "/rgba"])
    assert f.executed == int(g[name + "/executed"])


def test_golden_palettes(oracle, golden):
    g = golden["palettes"]
    ts = g["t"]
    for shader, name, modes in ((0, "mandelbrot", MANDEL_PALETTES), (1, "julia", JULIA_PALETTES)):
        for m in modes:
            cur = np.stack([oracle.palette(shader, m, float(t)) for t in ts])
            assert np.array_equal(cur, g["%s/%d" % (name, m)])


def test_workload_statistics_match_survey(oracle):
    """BASELINE.md section 4 workload table (measured during the survey with a throw-away script)."""
    f = oracle.render(oracle.OracleParams(), 256, 256, planes=True)
    assert abs(f.executed / 256 ** 2 - 48.5) < 1.0 and abs((f.iter == 256).mean() - 0.169) < 0.01
    f = oracle.render(oracle.OracleParams(max_iterations=1024), 256, 256)
    assert abs(f.executed / 256 ** 2 - 177.9) < 2.0


# ---- constants pinned by the reference's compiled shaders ---------------------------------------------
def _float_literals(text):
    import re
    import struct
    vals = set()
    for m in re.finditer(r"(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+)f\b", text):
        vals.add(struct.unpack("<I", struct.pack("<f", float(m.group(1))))[0])
    return vals


def _function_body(src, signature):
    i = src.rindex(signature)            # the definition, not a forward declaration
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        depth += {"{": 1, "}": -1}.get(src[k], 0)
        if depth == 0:
            return src[j:k + 1]
        k += 1


def test_palette_and_post_chain_literals_exist_in_the_reference_spirv():
    """tests/golden/spv_constants.json holds the float constants of the reference's compiled shaders
    (shaders/*.comp.spv; extracted by tests/golden/make_spv_constants.py).  Every float literal typed into the
    palette and post-chain code of the oracle AND of the library's knot-table builder must be one of them:
    a mistyped knot, break point, exponent or tonemap coefficient would not be."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spv = json.load(open(os.path.join(root, "tests", "golden", "spv_constants.json")))
    sets = {k: set(v["float32_bits"]) for k, v in spv.items()}
    oracle_c = open(os.path.join(root, "oracle", "fr_oracle.c")).read()
    host_c = open(os.path.join(root, "fractalrenderer_amd", "csrc", "fr_host.c")).read()
    # (smoothstep is a GLSL built-in, an extended instruction in SPIR-V: its 3 - 2t polynomial has no constants there)
    shared = _float_literals(_function_body(oracle_c, "static void ramp_quarters(")) | \
        _float_literals(_function_body(oracle_c, "static void ramp_fifths("))
    checks = [
        ("mandelbrot", _float_literals(_function_body(oracle_c, "static void palette_mandelbrot(")) | shared),
        ("julia", _float_literals(_function_body(oracle_c, "static void palette_julia(")) | shared),
        ("burning_ship", _float_literals(_function_body(oracle_c, "static void palette_julia(")) | shared),
        ("mandelbrot", _float_literals(_function_body(oracle_c, "static inline float aces("))),
        ("julia", _float_literals(_function_body(oracle_c, "static inline float aces("))),
        # enhance_color + gamma: the shader's 1.0 / 2.2 is folded to one constant in the binary (checked below)
        ("mandelbrot", _float_literals(_function_body(oracle_c, "void fro_post_chain(")) - _float_literals("2.2f")),
    ]
    # the library's own table builder (both shaders' numbering in one function)
    table = _float_literals(_function_body(host_c, "static void palette_table_fill(")) | \
        _float_literals(_function_body(host_c, "static void ramp_quarters(")) | \
        _float_literals(_function_body(host_c, "static void ramp_fifths("))
    assert len(table) > 20
    assert table <= (sets["mandelbrot"] | sets["julia"]), sorted(table - (sets["mandelbrot"] | sets["julia"]))
    for shader, lits in checks:
        assert lits, shader
        missing = lits - sets[shader]
        assert not missing, (shader, [__import__("struct").unpack("<f", __import__("struct").pack("<I", b))[0] for b in sorted(missing)])
    # folded constants the GLSL compiler left behind: log(2.0) and 1.0/2.2 as float32
    import struct
    f32 = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]  # noqa: E731
    assert f32(0.6931471805599453) in sets["mandelbrot"] and f32(1.0 / 2.2) in sets["mandelbrot"]
    assert f32(1e20) in sets["mandelbrot"] and f32(1e10) in sets["burning_ship"]      # min_trap / min_orbit_dist seeds
