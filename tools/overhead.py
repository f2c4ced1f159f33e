#!/usr/bin/env python3
"""Per-pixel fixed cost vs per-iteration cost: C2 view at 4096^2 for several max_iter and plane sets."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr

W = H = 4096
r = fr.Renderer(0)
dev = "cuda:0"
rgba = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
nu = torch.empty((H, W), dtype=torch.float64, device=dev)
it = torch.empty((H, W), dtype=torch.int32, device=dev)
def t(state, reps=5, **planes):
    ts = []
    for k in range(reps + 1):
        r.render(state, W, H, **planes)
        if k: ts.append(r.last_kernel_ms())
    return statistics.median(ts)
print("max_iter  rgba_ms  iter_ms  nu_ms   (C2 view, 4096^2, fp64)")
for mi in (1, 2, 16, 17, 32, 64, 128, 256, 512, 1024):
    st = fr.FractalState(max_iterations=mi)
    print(f"{mi:6d}  {t(st, rgba=rgba):8.4f} {t(st, iter=it):8.4f} {t(st, nu=nu):8.4f}")
print("far exterior view (every pixel escapes at i<=1), max_iter 1024:")
st = fr.FractalState(center_x=8.0, center_y=8.0, zoom=2.0, max_iterations=1024)
print(f"        {t(st, rgba=rgba):8.4f} {t(st, iter=it):8.4f} {t(st, nu=nu):8.4f}")
print("deep interior view (no pixel escapes), max_iter 1024:")
st = fr.FractalState(center_x=-0.2, center_y=0.0, zoom=0.2, max_iterations=1024)
print(f"        {t(st, rgba=rgba):8.4f} {t(st, iter=it):8.4f} {t(st, nu=nu):8.4f}   ideal at 6.18 T iter/s: {W*H*1024/6.176e12*1e3:.4f}")
