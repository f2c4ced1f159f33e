#!/usr/bin/env python3
"""Print-export sizes (src/vk_engine.cpp:1796-2232: export_print_quality(w, h, supersample)): sample grids above 2^29 samples.
The sample loop of the general tile kernel ("ssaa" = 1) against the automatic choice (staged, in bands of whole sub-tile rows
through one scratch)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
r = fr.Renderer(0)
for name, W, H, aa, kw in (("8192^2 aa 3, default view mi 1024 fp64", 8192, 8192, 3, dict(max_iterations=1024)),
                           ("8192^2 aa 3, seahorse 0.008 mi 2048 fp64", 8192, 8192, 3, dict(max_iterations=2048, center_x=-0.743643887037151, center_y=0.13182590420533, zoom=0.008)),
                           ("8192^2 aa 4, default view mi 1024 fp64", 8192, 8192, 4, dict(max_iterations=1024))):
    st = fr.FractalState(antialiasing_samples=aa, **kw)
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    ref = torch.empty_like(out)
    res = {}
    for mode in (1, 0):
        r.set_option("ssaa", mode)
        ts = []
        for k in range(3):
            r.render(st, W, H, rgba=(ref if mode == 1 else out))
            if k: ts.append(r.last_kernel_ms())
        res[mode] = statistics.median(ts)
    same = bool(torch.equal(ref, out))
    free, total = torch.cuda.mem_get_info()
    print(f"{name:44s}: sample loop {res[1]:9.2f} ms   automatic (bands) {res[0]:9.2f} ms ({100 * (res[0] / res[1] - 1):+.0f} %)   identical: {same}   device memory in use {(total - free) / 2**30:.1f} GiB", flush=True)
