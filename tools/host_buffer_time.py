#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer boundary (FR_MEM_HOST): C2 frame rendered into host memory."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fractalrenderer_amd as fr
W = H = 4096
r = fr.Renderer(0)
st = fr.FractalState(max_iterations=1024)
for name, buf in (("pageable numpy", np.empty((H, W, 4), np.float32)),
                  ("pinned (torch pin_memory)", torch.empty((H, W, 4), dtype=torch.float32, pin_memory=True).numpy())):
    ts = []
    for k in range(6):
        t0 = time.perf_counter()
        r.render(st, W, H, rgba=buf)
        ts.append(time.perf_counter() - t0)
    ms = statistics.median(ts[1:]) * 1e3
    print("%-28s %.2f ms/frame = %.0f Mpx/s (kernel %.3f ms; 256 MiB over PCIe: %.1f GB/s effective)" % (
        name, ms, W * H / ms / 1e3, r.last_kernel_ms(), 268.4 / (ms - r.last_kernel_ms())))
