#!/usr/bin/env python3
"""Two-pass (tile pass + lane pool) against single pass with home-shard probing, over frame size and max_iter."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
r = fr.Renderer(0)
views = {"default": dict(), "seahorse": dict(center_x=-0.743643887037151, center_y=0.13182590420533, zoom=0.008),
         "julia": dict(center_x=0.0, julia_c_real=-0.8, julia_c_imag=0.156)}
variants = {"auto": {}, "two-pass": {"staging": 3}, "single,probes=1": {"staging": 1, "probes": 1}}
print("view      prec   size        max_iter " + " ".join("%16s" % v for v in variants))
for vname, vkw in views.items():
    ft = fr.FractalType.JuliaSet if vname == "julia" else fr.FractalType.Mandelbrot
    for prec in (fr.Precision.F64, fr.Precision.F32):
        for W, H in ((1280, 720), (1920, 1080), (3840, 2160), (4096, 4096)):
            out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
            for mi in (128, 256, 384, 512, 768, 1024, 2048):
                st = fr.FractalState(max_iterations=mi, **vkw)
                res = []
                for v, opts in variants.items():
                    for k in ("staging", "probes"): r.set_option(k, opts.get(k, 0))
                    ts = []
                    for k in range(9):
                        r.render(st, W, H, fractal_type=ft, precision=prec, rgba=out)
                        if k: ts.append(r.last_kernel_ms())
                    res.append(statistics.median(ts))
                print("%-9s %-5s %5dx%-5d %7d  " % (vname, prec.name, W, H, mi) + " ".join("%16.4f" % t for t in res), flush=True)
            del out
