#!/usr/bin/env python3
"""Probe: is fr_render_shard_async capturable into a HIP graph after fr_ctx_reserve?  Prints what a replay writes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
W, H = 1000, 700
st = fr.FractalState(max_iterations=1024)
r = fr.Renderer(0)
r.reserve(st, W, H)
nu = torch.full((H, W), -1.0, dtype=torch.float64, device="cuda:0")
marker = torch.zeros(4, device="cuda:0")
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.graph(g, stream=side):
    marker += 1
    r.render(st, W, H, nu=nu, sync=False, stream=torch.cuda.current_stream().cuda_stream)
    marker += 10
torch.cuda.synchronize()
print("after capture: marker", marker.tolist(), "nu min/max", float(nu.min()), float(nu.max()))
for k in range(2):
    nu.fill_(-1.0)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print("replay", k, "marker", marker.tolist(), "nu min/max", float(nu.min()), float(nu.max()), "written", int((nu != -1).sum()))
try:
    r.check(); print("check ok")
except Exception as e:
    print("check:", e)
