#!/usr/bin/env python3
"""Does replaying a captured render (prepare + tile pass + lane pool as ONE hipGraph) shorten a frame?  The two ~5 us gaps
between the dependent launches are what a graph could close.  usage: graph_time.py [workload] [frames]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w = WORKLOADS[name]; W, H = w["W"], w["H"]
st = fr.FractalState(**w["state"])
kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]])
r = fr.Renderer(0)
r.set_option("periodicity", -1)
r.reserve(st, W, H, **kw)
rgba = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
side = torch.cuda.Stream()
h = side.cuda_stream
for _ in range(5):
    r.render(st, W, H, rgba=rgba, sync=False, stream=h, **kw)
torch.cuda.synchronize()
want = rgba.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    r.render(st, W, H, rgba=rgba, sync=False, stream=torch.cuda.current_stream().cuda_stream, **kw)
torch.cuda.synchronize()
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(side)
        for _ in range(K): fn()
        e1.record(side)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
direct, graph = [], []
for rd in range(7):
    direct.append(timed(lambda: r.render(st, W, H, rgba=rgba, sync=False, stream=h, **kw)))
    rgba.zero_()
    graph.append(timed(lambda: g.replay()))
    assert torch.equal(rgba, want), "a replayed render differs"
print(f"{name}: direct launches {statistics.median(direct):.4f} ms/frame (min {min(direct):.4f}), graph replay {statistics.median(graph):.4f} (min {min(graph):.4f}), {K} frames back to back x 7 rounds")
