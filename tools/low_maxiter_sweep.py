import os, sys, statistics, random
sys.path.insert(0, os.getcwd())
import torch, fractalrenderer_amd as fr
r = fr.Renderer(0)
cases = [("julia", dict(center_x=0.0, julia_c_real=-0.8, julia_c_imag=0.156), fr.FractalType.JuliaSet),
         ("default", dict(), fr.FractalType.Mandelbrot), ("seahorse", dict(center_x=-0.743643887037151, center_y=0.13182590420533, zoom=0.008), fr.FractalType.Mandelbrot)]
variants = ["", "stage_first=16", "stage_first=48", "stage_first=64", "pool_items_per_wg=16", "pool_items_per_wg=64", "pool_refill_at=8", "pool_refill_at=32"]
ALL = ("stage_first", "pool_items_per_wg", "pool_refill_at")
random.seed(3)
for name, kw, ft in cases:
    for prec in (fr.Precision.F64, fr.Precision.F32):
        for (W, H) in ((1920, 1080), (3840, 2160)):
            for mi in (256, 512):
                if ft == fr.FractalType.Mandelbrot and (mi < 512 or prec == fr.Precision.F32): continue
                st = fr.FractalState(max_iterations=mi, **kw)
                out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
                t = {v: [] for v in variants}
                for rd in range(12):
                    order = list(variants); random.shuffle(order)
                    for v in order:
                        for k in ALL: r.set_option(k, 0)
                        if v: r.set_option(v.split("=")[0], int(v.split("=")[1]))
                        r.render(st, W, H, fractal_type=ft, precision=prec, rgba=out)
                        if rd: t[v].append(r.last_kernel_ms())
                base = statistics.median(t[""])
                print(f"{name:8s} {prec.name} {W}x{H} mi {mi} stages {r.last_stages()}: default {base:.4f} | " + "  ".join(f"{v} {100*(statistics.median(t[v])/base-1):+.1f}%" for v in variants[1:]), flush=True)
