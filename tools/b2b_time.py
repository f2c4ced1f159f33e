#!/usr/bin/env python3
"""Back-to-back frames as bench.py times them (no event pair per render, one synchronisation per batch), for A/B runs of
library builds: FR_LIB_PATH=build/ab/x.so tools/b2b_time.py frames reps workload[:k=v,k=v] ..."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
frames, reps = int(sys.argv[1]), int(sys.argv[2])
for spec in sys.argv[3:]:
    name, _, opts = spec.partition(":")
    w = WORKLOADS[name]; W, H = w["W"], w["H"]
    state = fr.FractalState(**w["state"])
    r = fr.Renderer(0, timing=False)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("="); r.set_option(k, int(v, 0))
    s = torch.cuda.Stream()
    kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]],
              rgba=torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0"), sync=False, stream=s.cuda_stream)
    ts = []
    for rep in range(reps + 2):
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames): r.render(state, W, H, **kw)
        s.synchronize()
        if rep >= 2: ts.append((time.perf_counter() - t0) / frames * 1e3)
    r.check()
    print(f"{spec:28s} {statistics.median(ts):.4f} ms (min {min(ts):.4f})  {os.path.basename(os.environ.get('FR_LIB_PATH', 'in-tree'))}", flush=True)
    r.close()
