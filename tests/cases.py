"""Named parity cases shared by the golden generator and the tests.

Each case = (oracle parameters, width, height).  Views follow BASELINE.md section 3
(C1..C5), at sizes the CPU oracle finishes in well under a second.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.oracle import OracleParams  # noqa: E402

SEAHORSE = (-0.743643887037151, 0.13182590420533)   # src/fractal_state.h:175, src/deep_zoom_system.cpp:576-583

CASES = {
    # C1: Mandelbrot default viewport (src/fractal_state.h:18-21), fp64
    "c1_mandel_f64_default": (OracleParams(), 64, 64),
    # C2 view at an odd, non-multiple-of-8 size (ragged edges in every sub-tile shape)
    "c2_mandel_f64_mi1024_ragged": (OracleParams(max_iterations=1024), 131, 67),
    # the reference's own precision on the same view
    "c2_mandel_f32_mi1024": (OracleParams(max_iterations=1024, precision=0), 96, 64),
    # FractalState::reset() view (zoom 1.5), 60 % interior
    "reset_view_f64": (OracleParams(zoom=1.5, max_iterations=512), 80, 48),
    # C3: Julia c = -0.8 + 0.156i, fp32, centre (0,0) and the reference's shared default centre
    "c3_julia_f32_centre0": (OracleParams(fractal=1, precision=0, center_x=0.0, center_y=0.0, zoom=3.0,
                                          max_iterations=2048, julia_c_real=-0.8, julia_c_imag=0.156), 96, 64),
    "c3_julia_f32_default_centre": (OracleParams(fractal=1, precision=0, zoom=3.0, max_iterations=2048,
                                                 julia_c_real=-0.8, julia_c_imag=0.156), 72, 40),
    "julia_f64_default_c": (OracleParams(fractal=1, precision=1, center_x=0.0, max_iterations=512), 64, 48),
    # C4: Seahorse deep preset, zoom 1e-6, max_iter 16384 (74.7 % interior)
    "c4_seahorse_deep_f64": (OracleParams(center_x=SEAHORSE[0], center_y=SEAHORSE[1], zoom=1e-6,
                                          max_iterations=16384), 40, 24),
    "seahorse_0008_f64": (OracleParams(center_x=SEAHORSE[0], center_y=SEAHORSE[1], zoom=0.008,
                                       max_iterations=1024), 64, 64),
    # colouring variants of a1
    "mandel_interior_black": (OracleParams(interior_style=1, max_iterations=128), 48, 48),
    "mandel_interior_trap": (OracleParams(interior_style=2, max_iterations=128, palette_mode=3), 48, 48),
    "mandel_trap_blend": (OracleParams(orbit_trap_enabled=1, orbit_trap_radius=0.35, max_iterations=128), 48, 48),
    "mandel_stripes_f32": (OracleParams(stripe_enabled=1, stripe_density=7.0, max_iterations=128, precision=0), 48, 48),
    "mandel_trap_stripes_f64": (OracleParams(orbit_trap_enabled=1, stripe_enabled=1, max_iterations=200,
                                             color_offset=0.25, color_scale=3.0, palette_mode=5), 56, 40),
    "mandel_aa2_post": (OracleParams(aa=2, post_chain=1, max_iterations=128, brightness=1.2, saturation=0.8,
                                     contrast=1.1, palette_mode=1), 40, 32),
    "julia_aa3_post_f32": (OracleParams(fractal=1, precision=0, center_x=0.0, aa=3, post_chain=1, max_iterations=256,
                                        palette_mode=4, brightness=0.05, contrast=0.05, saturation=-1.0), 40, 32),
    "mandel_scale_offset": (OracleParams(color_scale=7.5, color_offset=0.37, palette_mode=4, max_iterations=300), 64, 32),
    # bailout outside the "absorbing escape" fast path (bailout^2 < 4.5) and a huge one
    "mandel_small_bailout": (OracleParams(bailout=1.5, max_iterations=200), 64, 48),
    "mandel_big_bailout": (OracleParams(bailout=1e4, max_iterations=300), 64, 48),
    "julia_c_outside_bailout": (OracleParams(fractal=1, center_x=0.0, julia_c_real=3.0, julia_c_imag=-2.5,
                                             bailout=3.0, max_iterations=100, zoom=12.0), 64, 48),
    # view far from the set: every pixel escapes within a few iterations
    "mandel_far_exterior": (OracleParams(center_x=5.0, center_y=5.0, zoom=2.0, max_iterations=64), 40, 40),
    # Burning Ship (shaders/burning_ship.comp): same loop with z = abs(z) before the square
    "ship_f64_overview": (OracleParams(fractal=2, center_x=-0.5, center_y=-0.5, zoom=3.5, max_iterations=256), 96, 64),
    "ship_f32_the_ship": (OracleParams(fractal=2, precision=0, center_x=-1.755, center_y=-0.03, zoom=0.08,
                                       max_iterations=1024, palette_mode=2), 80, 48),
    "ship_f64_ragged_mi2048": (OracleParams(fractal=2, center_x=-1.76, center_y=-0.02, zoom=0.12, max_iterations=2048,
                                            color_scale=4.0, palette_mode=6), 131, 67),
    "ship_trap_style1_f64": (OracleParams(fractal=2, center_x=-0.5, center_y=-0.5, zoom=3.5, max_iterations=160,
                                          orbit_trap_enabled=1, orbit_trap_radius=0.6, interior_style=1,
                                          palette_mode=3), 64, 48),
    "ship_stripes_style2_f32": (OracleParams(fractal=2, precision=0, center_x=-0.5, center_y=-0.5, zoom=3.5,
                                             max_iterations=96, stripe_enabled=1, stripe_density=7.0,
                                             interior_style=2, palette_mode=1), 64, 48),
    "ship_style3_aa2_post": (OracleParams(fractal=2, center_x=-0.5, center_y=-0.5, zoom=3.5, max_iterations=128,
                                          interior_style=3, aa=2, post_chain=1, brightness=1.15, contrast=1.3,
                                          saturation=0.7, palette_mode=8), 40, 32),
    "ship_small_bailout_f64": (OracleParams(fractal=2, center_x=-0.5, center_y=-0.5, zoom=3.5, max_iterations=200,
                                            bailout=1.5), 64, 48),
    # Deep_Zoom: the reference's perturbation shader (shaders/test_deep_zoom.comp), fp32 float-float
    "deepzoom_seahorse": (OracleParams(fractal=5, precision=0, center_x=SEAHORSE[0], center_y=SEAHORSE[1], zoom=1e-6,
                                       max_iterations=2000, use_perturbation=1), 72, 40),
    "deepzoom_wide_escaping_reference": (OracleParams(fractal=5, precision=0, center_x=-0.75, center_y=0.1, zoom=100.0,
                                                      max_iterations=300, use_perturbation=1, palette_mode=1,
                                                      color_scale=2.0, color_offset=0.5), 64, 64),
    "deepzoom_no_perturbation": (OracleParams(fractal=5, precision=0, center_x=-0.6, center_y=0.2, zoom=150.0,
                                              max_iterations=200, use_perturbation=0, palette_mode=2), 56, 40),
    "deepzoom_gray_small_bailout": (OracleParams(fractal=5, precision=0, center_x=-0.1, center_y=0.65, zoom=120.0,
                                                 max_iterations=150, use_perturbation=1, palette_mode=7, bailout=1.0), 48, 48),
    # tiny frames
    "mandel_1x1": (OracleParams(), 1, 1),
    "mandel_3x70": (OracleParams(max_iterations=100), 3, 70),
}

MANDEL_PALETTES = list(range(-1, 8))     # 0..5 defined, others fall back to fire (shaders/mandelbrot.comp:139)
JULIA_PALETTES = list(range(-1, 12))     # 0..9 defined, others fall back to ultra_fire (shaders/julia.comp:178)


def needs_effects(p) -> bool:
    """Colourings that use more of the orbit than (escape index, |z|^2): rendered by the as-written
    effects variant of the tile kernel, never staged."""
    if p.fractal == 0:
        return bool(p.orbit_trap_enabled or p.stripe_enabled or p.interior_style == 2)
    if p.fractal == 2:
        return bool(p.orbit_trap_enabled or (p.stripe_enabled and p.interior_style == 2) or p.interior_style == 3)
    return False
