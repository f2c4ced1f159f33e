#!/usr/bin/env python3
"""render -> 8-bit export on one stream, back to back (what an animation export does per frame), for A/B runs of library
builds: FR_LIB_PATH=... tools/render_export_time.py frames reps workload ..."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
frames, reps = int(sys.argv[1]), int(sys.argv[2])
for name in sys.argv[3:]:
    w = WORKLOADS[name]; W, H = w["W"], w["H"]
    state = fr.FractalState(**w["state"])
    r = fr.Renderer(0, timing=False)
    s = torch.cuda.Stream()
    rgba = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
    kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]], post_chain=True, rgba=rgba, sync=False, stream=s.cuda_stream)
    ts = []
    for rep in range(reps + 2):
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            r.render(state, W, H, **kw)
            r.export_rgb8(rgba, W, H, out=rgb8, through_half=True, stream=s.cuda_stream)
        s.synchronize()
        if rep >= 2: ts.append((time.perf_counter() - t0) / frames * 1e3)
    print(f"{name:10s} render + export8 {statistics.median(ts):.4f} ms (min {min(ts):.4f})  {os.path.basename(os.environ.get('FR_LIB_PATH', 'in-tree'))}", flush=True)
    r.close()
