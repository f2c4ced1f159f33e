#!/usr/bin/env python3
"""Per-wave timeline of one render from the kernel's diag buffer: when waves start/finish, how the
work and the dequeues are spread.  usage: timeline.py workload [max_iter] [opt=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
w = WORKLOADS[sys.argv[1]]
st = dict(w["state"])
args = sys.argv[2:]
if args and args[0].isdigit():
    st["max_iterations"] = int(args[0]); args = args[1:]
W, H = w["W"], w["H"]
r = fr.Renderer(0)
for a in args:
    k, v = a.split("="); r.set_option(k, int(v))
rgba = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]], rgba=rgba)
state = fr.FractalState(**st)
r.render(state, W, H, **kw)
nw = r.last_grid() * 4
diag = torch.zeros((nw, 4), dtype=torch.int64, device="cuda:0")
r.set_option("diag_buffer", diag.data_ptr())
r.render(state, W, H, **kw); ms = r.last_kernel_ms()
r.render(state, W, H, **kw); ms = r.last_kernel_ms()
d = diag.cpu().numpy()
ran = d[:, 1] > 0
t0 = d[ran, 0].min()
start = (d[ran, 0] - t0) / 100.0   # us (100 MHz ticks)
end = (d[ran, 1] - t0) / 100.0
sub, claims = d[ran, 2], d[ran, 3]
span = end.max()
print(f"{sys.argv[1]} {st}: kernel {ms:.4f} ms, grid {nw//4} WG, waves that ran {ran.sum()}/{nw}, span {span:.1f} us")
print("  wave start  us  p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(start, [50, 90, 99, 100])))
print("  wave finish us  p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(end, [1, 10, 50, 90, 99, 100])))
worked = sub > 0
print(f"  waves with work {worked.sum()}, sub-tiles/wave mean {sub[worked].mean():.1f} min {sub[worked].min()} max {sub[worked].max()}, "
      f"dequeues total {claims.sum()} (mean/wave {claims.mean():.1f}), sub-tiles/dequeue {sub.sum()/max(1,claims.sum()-8*ran.sum()):.2f}")
# how much SIMD-time is idle at the end: integrate number of live waves over time
edges = np.linspace(0, span, 41)
live = [(np.minimum(end, b) - np.maximum(start, a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("  live waves per 2.5% time slice:", " ".join(f"{x:.0f}" for x in live))
