#!/usr/bin/env python3
"""Supersampled frames with stripe shading: the sample loop of the effects variant ("stripes" = 1) against the staged sample
grid in the lean kernels' stripe instantiations (automatic), interleaved."""
import os, random, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
r = fr.Renderer(0)
random.seed(3)
for name, W, H, kw in (("default view 4096^2 mi 1024 fp64", 4096, 4096, dict(max_iterations=1024)),
                       ("seahorse 0.008 4096^2 mi 2048 fp64", 4096, 4096, dict(max_iterations=2048, center_x=-0.743643887037151, center_y=0.13182590420533, zoom=0.008)),
                       ("1080p mi 512 fp64", 1920, 1080, dict(max_iterations=512))):
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    ref = torch.empty_like(out)
    for aa in (1, 2, 3):
        st = fr.FractalState(stripe_enabled=True, antialiasing_samples=aa, **kw)
        t = {1: [], 0: []}
        for rd in range(6):
            order = [1, 0]; random.shuffle(order)
            for m in order:
                r.set_option("stripes", m)
                r.render(st, W, H, rgba=(ref if m else out))
                if rd: t[m].append(r.last_kernel_ms())
        a, b = statistics.median(t[1]), statistics.median(t[0])
        print(f"{name:36s} aa={aa}: effects variant {a:8.3f} ms   lean stripe instantiations {b:8.3f} ms ({100 * (b / a - 1):+.0f} %)   identical: {bool(torch.equal(ref, out))}", flush=True)
