#!/usr/bin/env python3
"""Small frames: one pass against tile pass + lane pool against the automatic choice, over size and max_iter (default view)."""
import os, random, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 15
r = fr.Renderer(0)
random.seed(4)
views = {"default": dict(), "seahorse": dict(center_x=-0.743643887037151, center_y=0.13182590420533, zoom=0.008),
         "julia": dict(center_x=0.0, julia_c_real=-0.8, julia_c_imag=0.156), "rabbit": dict(center_x=0.0, julia_c_real=-0.123, julia_c_imag=0.745)}
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(views)
for vname, vkw in views.items():
    if vname not in only: continue
    ft = fr.FractalType.JuliaSet if "julia_c_real" in vkw else fr.FractalType.Mandelbrot
    for W, H in ((256, 256), (400, 300), (512, 512), (640, 480), (800, 600), (1024, 768)):
        out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
        for prec in (fr.Precision.F64, fr.Precision.F32):
            for mi in ((256, 512, 1024, 2048) if "julia_c_real" in vkw else (512, 1024, 2048, 4096)):
                st = fr.FractalState(max_iterations=mi, **vkw)
                t = {0: [], 1: [], 3: []}
                for rd in range(rounds + 1):
                    order = [0, 1, 3]; random.shuffle(order)
                    for m in order:
                        r.set_option("staging", m)
                        r.render(st, W, H, fractal_type=ft, precision=prec, rgba=out)
                        if rd: t[m].append(r.last_kernel_ms())
                a, b, c = (statistics.median(t[k]) * 1e3 for k in (0, 1, 3))
                flag = "  <== automatic not best" if a > 1.05 * min(b, c) else ""
                print(f"{vname:9s} {W:5d}x{H:<5d} {prec.name} mi {mi:5d}: automatic {a:7.1f} us   one pass {b:7.1f}   two passes {c:7.1f}{flag}", flush=True)
