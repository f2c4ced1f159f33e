#!/usr/bin/env python3
"""SSAA: the sample loop of the general tile kernel ("ssaa" = 1) against the staged form ("ssaa" = 2: sample grid through the lean
tile pass + lane pool, then ssaa_reduce_kernel) and the automatic choice, interleaved.  usage: ssaa_ab.py [periodicity]"""
import os, random, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
r = fr.Renderer(0)
per = int(sys.argv[1]) if len(sys.argv) > 1 else 0
r.set_option("periodicity", per)
print("periodicity", per)
random.seed(5)
for name, ft, prec, W, H, kw in (
        ("c2 4096^2 mi 1024 fp64", fr.FractalType.Mandelbrot, fr.Precision.F64, 4096, 4096, dict(max_iterations=1024)),
        ("c3 4096^2 mi 2048 fp32 julia", fr.FractalType.JuliaSet, fr.Precision.F32, 4096, 4096, dict(max_iterations=2048, center_x=0.0, julia_c_real=-0.8, julia_c_imag=0.156)),
        ("c5 view 4096^2 mi 4096 fp64", fr.FractalType.Mandelbrot, fr.Precision.F64, 4096, 4096, dict(max_iterations=4096, center_x=-0.743643887037151, center_y=0.13182590420533, zoom=0.008)),
        ("1080p mi 256 fp32 (the reference's draw loop)", fr.FractalType.Mandelbrot, fr.Precision.F32, 1920, 1080, dict(max_iterations=256)),
        ("1080p mi 1024 fp64", fr.FractalType.Mandelbrot, fr.Precision.F64, 1920, 1080, dict(max_iterations=1024))):
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    for aa in (2, 3):
        st = fr.FractalState(antialiasing_samples=aa, **kw)
        t = {1: [], 2: [], 0: []}
        for rd in range(6):
            order = [1, 2, 0]; random.shuffle(order)
            for m in order:
                r.set_option("ssaa", m)
                r.render(st, W, H, fractal_type=ft, precision=prec, rgba=out)
                if rd: t[m].append(r.last_kernel_ms())
        a, b, c = (statistics.median(t[m]) for m in (1, 2, 0))
        print(f"{name:46s} aa={aa}: sample loop {a:8.3f} ms   staged {b:8.3f} ms ({100 * (b / a - 1):+.0f} %)   automatic {c:8.3f} ms", flush=True)
