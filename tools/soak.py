#!/usr/bin/env python3
"""Longer randomised parity soak (not part of the test suite): usage soak.py [trials] [seed] [size scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fractalrenderer_amd as fr
from oracle import oracle
import test_gpu_parity as T
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1        # frame sizes up to 420*scale x 300*scale
rng = np.random.default_rng(seed)
r = fr.Renderer(0)
anchors = {0: [(-0.743643887037151, 0.13182590420533), (-0.1011, 0.9563), (-1.25066, 0.02012), (0.275, 0.0), (-0.5, 0.0), (-1.7497, 0.00001)],
           1: [(0.0, 0.0), (0.3, 0.2), (-0.6, 0.1)], 2: [(-1.755, -0.03), (-0.5, -0.5), (-1.62, -0.002)]}
opts = ["staging", "pool_refill_at", "probes", "stream_probes", "stream_rotate", "stage_first", "subtile_shape", "workgroups_per_cu", "periodicity",
        "tile_kernel", "tile_pixels", "tile_exit", "tile_exit_from", "prepare", "stripes", "shards", "regions"]
bad = 0
for trial in range(trials):
    fractal = int(rng.integers(0, 3)); prec = int(rng.integers(0, 2))
    ax, ay = anchors[fractal][int(rng.integers(0, len(anchors[fractal])))]
    zoom = float(10.0 ** rng.uniform(-5.0 if prec == 1 else -2.0, 0.6))
    W, H = int(rng.integers(9, 420 * scale)), int(rng.integers(5, 300 * scale))
    kw = dict(fractal=fractal, precision=prec, center_x=ax + zoom * float(rng.uniform(-0.2, 0.2)), center_y=ay + zoom * float(rng.uniform(-0.2, 0.2)),
              zoom=zoom, max_iterations=int(rng.choice([1, 33, 127, 128, 129, 300, 777, 1500, 3000, 6000])),
              bailout=float(rng.choice([1.5, 2.0, 2.5, 4.0, 4.0, 16.0, 1000.0])), palette_mode=int(rng.integers(0, 6 if fractal == 0 else 10)),
              color_offset=float(np.float32(rng.uniform(0, 1))), color_scale=float(np.float32(rng.uniform(0.5, 6))),
              interior_style=int(rng.choice([0, 0, 1])), post_chain=int(rng.integers(0, 2)), aa=int(rng.choice([1, 1, 1, 2])))
    if fractal == 1:
        kw.update(julia_c_real=float(rng.uniform(-0.9, 0.4)), julia_c_imag=float(rng.uniform(-0.7, 0.7)))
    if prec == 1 and fractal in (0, 2) and rng.random() < 0.3:
        # the effects variants (orbit trap, stripes, interior styles): fp64 only -- in fp32 the stripe angle inherits
        # the escape z's last bits
        kw.update(orbit_trap_enabled=int(rng.integers(0, 2)), stripe_enabled=int(rng.integers(0, 2)),
                  interior_style=int(rng.choice([0, 1, 2, 3])), orbit_trap_radius=float(np.float32(rng.uniform(0.1, 1.5))),
                  stripe_density=float(np.float32(rng.uniform(1.0, 20.0))))
        if rng.random() < 0.5:       # stripes alone: the lean kernels' stripe instantiations (Mandelbrot)
            kw.update(orbit_trap_enabled=0, stripe_enabled=1, interior_style=int(rng.choice([0, 0, 1])))
    if prec == 0:
        kw["aa"] = 1          # fp32 samples next to the palette's fract() wrap legitimately flip; only checkable per pixel
    p = oracle.OracleParams(**kw)
    ref = oracle.render(p, W, H)
    tune = {}
    if trial % 2:
        tune = {"staging": int(rng.choice([0, 1, 3])), "pool_refill_at": int(rng.choice([0, 1, 8, 40, 64])),
                "probes": int(rng.choice([0, 1, 2, 8])), "stream_probes": int(rng.choice([0, 1, 4, 8])), "stream_rotate": int(rng.choice([0, 1, 2])),
                "stage_first": int(rng.choice([0, 16, 48, 160])), "subtile_shape": int(rng.choice([0, 3, 4, 6])), "workgroups_per_cu": int(rng.choice([0, 1, 3, 7]))}
    tune["periodicity"] = int(rng.choice([-1, 0, 1, 16, 64, 1000]))            # exact cycle closing: never changes a pixel
    # lean / general tile kernel, one or two sub-tiles per trip, 8 or 64 queue shards and stream regions
    tune.update(tile_kernel=int(rng.choice([0, 0, 1])), tile_pixels=int(rng.choice([0, 1, 2])), shards=int(rng.choice([0, 8, 64])),
                regions=int(rng.choice([0, 0, 8, 64])))
    # occupancy exit of the lean tile pass: off, automatic, "as soon as one sample has finished", never
    tune.update(tile_exit=int(rng.choice([0, 0, 1, 2, 8, 4096])), tile_exit_from=int(rng.choice([0, 0, 1, 16, 64])))
    tune["prepare"] = int(rng.choice([0, 0, 0, 1]))          # the tile pass's own prologue (automatic) / prepare_kernel in front
    tune["stripes"] = int(rng.choice([0, 0, 0, 1]))          # stripe shading through the lean kernels (automatic) / the effects variant
    for k in opts: r.set_option(k, tune.get(k, 0))
    shard = None
    if trial % 3 == 1:
        n = int(rng.integers(2, 9))
        shard = fr.Shard(int(rng.integers(0, n)), n, int(rng.choice([8, 16, 24, 32])) if rng.random() < 0.5 else int(rng.integers(1, 40)))
    if shard is not None and shard.rows(H) == 0:
        continue
    try:
        rgba, nu, it = T.gpu_render(fr, r, p, W, H, shard=shard)
    except fr.FractalRendererError as e:          # e.g. the lane pool's watchdog (FR_ERR_INTERNAL)
        bad += 1
        print("ERROR trial", trial, kw, W, H, shard, tune, e, flush=True)
        continue
    rows = shard.global_rows(H) if shard else slice(None)
    try:
        T.check_against(p, ref.iter[rows], ref.nu[rows], ref.rgba[rows], rgba, nu, it)
    except AssertionError as e:
        bad += 1
        print("FAIL trial", trial, kw, W, H, shard, tune, e, flush=True)
    if trial % 25 == 24: print("trial", trial + 1, "failures", bad, flush=True)
print("done: %d trials, %d failures" % (trials, bad))
sys.exit(1 if bad else 0)
