"""Cases whose expected values come from EXECUTING the reference's compiled shaders.

tests/golden/make_spv_golden.py runs shaders/<shader>.comp.spv of the reference through the SPIR-V
interpreter in tests/golden/spirv_interp.py, one invocation per pixel, with push constants packed from
these parameters, and stores what each invocation writes to the image (plus the escape index / smooth
value of its sample function, read through the OpName debug names) in tests/golden/spv_frames.npz.
All of them are fp32 cases (precision=0): that is the arithmetic the shaders compute in.
"""
from oracle.oracle import OracleParams

SEAHORSE = (-0.743643887037151, 0.13182590420533)

# name -> (shader, OracleParams, W, H)
SPV_CASES = {}


def _add(name, shader, W, H, **kw):
    fractal = {"mandelbrot": 0, "julia": 1, "burning_ship": 2, "test_deep_zoom": 5}[shader]
    kw.setdefault("post_chain", 0 if fractal == 5 else 1)
    SPV_CASES[name] = (shader, OracleParams(fractal=fractal, precision=0, **kw), W, H)


# ---- shaders/mandelbrot.comp.spv ---------------------------------------------------------------------------
_add("m_default", "mandelbrot", 64, 48, max_iterations=256)
for _m in range(6):
    _add("m_palette%d" % _m, "mandelbrot", 16, 12, max_iterations=64, palette_mode=_m, color_offset=0.1, color_scale=1.7)
_add("m_seahorse", "mandelbrot", 24, 16, center_x=SEAHORSE[0], center_y=SEAHORSE[1], zoom=0.004, max_iterations=256,
     palette_mode=3)
_add("m_trap", "mandelbrot", 24, 16, max_iterations=96, orbit_trap_enabled=1, orbit_trap_radius=0.5, palette_mode=1)
_add("m_stripe", "mandelbrot", 24, 16, max_iterations=96, stripe_enabled=1, stripe_density=10.0, palette_mode=4)
_add("m_trap_stripe_aa2", "mandelbrot", 12, 8, max_iterations=64, orbit_trap_enabled=1, orbit_trap_radius=0.75,
     stripe_enabled=1, stripe_density=6.5, aa=2, palette_mode=5)
for _s in (1, 2, 3):
    _add("m_interior%d" % _s, "mandelbrot", 20, 14, max_iterations=48, interior_style=_s, orbit_trap_enabled=_s == 3)
_add("m_aa2", "mandelbrot", 12, 8, max_iterations=64, aa=2)
_add("m_aa3", "mandelbrot", 8, 6, max_iterations=48, aa=3, palette_mode=2)
_add("m_post", "mandelbrot", 16, 12, max_iterations=64, brightness=1.2, saturation=0.8, contrast=1.1, palette_mode=1)
_add("m_bailout8_ragged", "mandelbrot", 17, 9, max_iterations=80, bailout=8.0, color_scale=0.5)
_add("m_bailout_small", "mandelbrot", 16, 12, max_iterations=64, bailout=1.5)

# out-of-range selectors: the shader's switch falls back to its default branch
_add("m_palette7_fallback", "mandelbrot", 16, 12, max_iterations=64, palette_mode=7, color_offset=0.1, color_scale=1.7)
_add("m_interior5_fallback", "mandelbrot", 16, 12, max_iterations=48, interior_style=5)
_add("m_aa4_offset", "mandelbrot", 6, 4, max_iterations=40, aa=4, color_offset=0.6, color_scale=3.0, palette_mode=3)
_add("m_wide_frame", "mandelbrot", 40, 6, center_x=-0.75, center_y=0.1, zoom=1.2, max_iterations=200, palette_mode=4)

# long orbits next to the boundary: 2000 chaotic updates amplify any difference in operation order or rounding
_add("m_long_orbits", "mandelbrot", 12, 8, center_x=SEAHORSE[0], center_y=SEAHORSE[1], zoom=0.0004, max_iterations=2000,
     palette_mode=1, color_scale=3.0)

# ---- shaders/julia.comp.spv --------------------------------------------------------------------------------
_add("j_default", "julia", 64, 48, center_x=0.0, max_iterations=256)
for _m in range(10):
    _add("j_palette%d" % _m, "julia", 16, 12, center_x=0.0, max_iterations=64, palette_mode=_m, color_offset=0.15,
         color_scale=1.3)
_add("j_dendrite_aa2", "julia", 12, 8, center_x=0.0, max_iterations=96, julia_c_real=0.0, julia_c_imag=1.0, aa=2)
_add("j_rabbit_post", "julia", 20, 14, center_x=0.0, zoom=2.5, max_iterations=96, julia_c_real=-0.123, julia_c_imag=0.745,
     brightness=1.1, saturation=1.3, contrast=0.9, palette_mode=4)
_add("j_bailout8", "julia", 16, 12, center_x=0.1, center_y=-0.2, zoom=1.5, max_iterations=80, bailout=8.0, palette_mode=6)

_add("j_palette12_fallback", "julia", 16, 12, center_x=0.0, max_iterations=64, palette_mode=12, color_offset=0.15, color_scale=1.3)
_add("j_aa3", "julia", 8, 6, center_x=0.0, max_iterations=64, aa=3, julia_c_real=0.285, julia_c_imag=0.01, palette_mode=2)
_add("j_tall_frame", "julia", 7, 33, center_x=0.0, zoom=2.8, max_iterations=150, julia_c_real=-0.4, julia_c_imag=0.6, palette_mode=5)

_add("j_long_orbits", "julia", 12, 8, center_x=1.2, center_y=0.0, zoom=0.01, max_iterations=2000,
     julia_c_real=-0.8, julia_c_imag=0.156, palette_mode=3)

# ---- shaders/burning_ship.comp.spv -------------------------------------------------------------------------
_add("s_default", "burning_ship", 64, 48, center_x=-0.5, center_y=-0.5, max_iterations=256)
for _m in range(10):
    _add("s_palette%d" % _m, "burning_ship", 16, 12, center_x=-0.5, center_y=-0.5, max_iterations=64, palette_mode=_m,
         color_offset=0.2, color_scale=1.5)
_add("s_ship_zoom", "burning_ship", 24, 16, center_x=-1.755, center_y=-0.03, zoom=0.08, max_iterations=192, palette_mode=8)
_add("s_trap", "burning_ship", 20, 14, center_x=-0.5, center_y=-0.5, max_iterations=64, orbit_trap_enabled=1,
     orbit_trap_radius=0.6, palette_mode=2)
_add("s_stripe", "burning_ship", 20, 14, center_x=-0.5, center_y=-0.5, max_iterations=64, stripe_enabled=1,
     stripe_density=7.0, palette_mode=5)
for _s in (1, 2, 3):
    _add("s_interior%d" % _s, "burning_ship", 20, 14, center_x=-0.5, center_y=-0.5, max_iterations=48, interior_style=_s,
         orbit_trap_enabled=_s == 3, stripe_enabled=_s == 2)
_add("s_aa2_post", "burning_ship", 12, 8, center_x=-0.5, center_y=-0.5, max_iterations=64, aa=2, brightness=1.1,
     saturation=0.9, contrast=1.2, palette_mode=9)

_add("s_palette11_fallback", "burning_ship", 16, 12, center_x=-0.5, center_y=-0.5, max_iterations=64, palette_mode=11)
_add("s_bailout8", "burning_ship", 18, 12, center_x=-1.755, center_y=-0.03, zoom=0.3, max_iterations=100, bailout=8.0, palette_mode=6)
_add("s_aa3_trap", "burning_ship", 8, 6, center_x=-0.5, center_y=-0.5, max_iterations=48, aa=3, orbit_trap_enabled=1, orbit_trap_radius=0.4)

_add("s_long_orbits", "burning_ship", 12, 8, center_x=-1.7720666666666665, center_y=-0.017733333333333334, zoom=0.0006, max_iterations=2000,
     palette_mode=7)

# ---- shaders/test_deep_zoom.comp.spv (FractalType::Deep_Zoom) ----------------------------------------------
_add("d_seahorse", "test_deep_zoom", 18, 10, center_x=SEAHORSE[0], center_y=SEAHORSE[1], zoom=1e-6, max_iterations=1200,
     use_perturbation=1)
_add("d_seahorse_mid", "test_deep_zoom", 20, 12, center_x=SEAHORSE[0], center_y=SEAHORSE[1], zoom=0.5, max_iterations=300,
     use_perturbation=1, palette_mode=0, color_scale=1.5)
_add("d_wide_escaping_reference", "test_deep_zoom", 16, 16, center_x=-0.75, center_y=0.1, zoom=100.0, max_iterations=120,
     use_perturbation=1, palette_mode=1, color_scale=2.0, color_offset=0.5)
_add("d_no_perturbation", "test_deep_zoom", 14, 10, center_x=-0.6, center_y=0.2, zoom=150.0, max_iterations=100,
     use_perturbation=0, palette_mode=2)
_add("d_gray_small_bailout", "test_deep_zoom", 12, 12, center_x=-0.1, center_y=0.65, zoom=120.0, max_iterations=90,
     use_perturbation=1, palette_mode=7, bailout=1.0)
_add("d_palette3_bailout4", "test_deep_zoom", 14, 10, center_x=-0.75, center_y=0.1, zoom=60.0, max_iterations=150,
     use_perturbation=1, palette_mode=3, bailout=4.0, color_scale=0.7)
