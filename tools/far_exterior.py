#!/usr/bin/env python3
"""Renders the far-exterior view (every pixel escapes at i <= 1) a few times: the tile pass's fixed per-pixel cost.
usage: far_exterior.py [plane=iter|nu|rgba] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
plane = sys.argv[1] if len(sys.argv) > 1 else "iter"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W = H = 4096
r = fr.Renderer(0)
bufs = {"rgba": torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0"),
        "nu": torch.empty((H, W), dtype=torch.float64, device="cuda:0"),
        "iter": torch.empty((H, W), dtype=torch.int32, device="cuda:0")}
st = fr.FractalState(center_x=8.0, center_y=8.0, zoom=2.0, max_iterations=1024)
for _ in range(reps):
    r.render(st, W, H, **{plane: bufs[plane]})
print("kernel ms", r.last_kernel_ms())
