#!/usr/bin/env python3
"""Deep_Zoom (the reference's perturbation shader) frame times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
r = fr.Renderer(0)
for W, H, zoom, mi in ((1920, 1080, 1e-6, 2000), (1920, 1080, 1e-3, 5000), (4096, 4096, 1e-6, 2000)):
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    st = fr.FractalState(center_x=-0.743643887037151, center_y=0.13182590420533, zoom=zoom, max_iterations=mi, use_perturbation=True)
    for _ in range(3):
        r.render(st, W, H, fractal_type=fr.FractalType.Deep_Zoom, precision=fr.Precision.F32, rgba=out)
    it = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
    r.render(st, W, H, fractal_type=fr.FractalType.Deep_Zoom, precision=fr.Precision.F32, rgba=out, iter=it)
    print("%dx%d zoom %g max_iter %d: kernel %.3f ms (%.0f Mpx/s), interior %.2f, mean iter %.0f" % (
        W, H, zoom, mi, r.last_kernel_ms(), W * H / r.last_kernel_ms() / 1e3, float((it >= mi).float().mean()), float(it.float().mean())), flush=True)
