#!/usr/bin/env python3
"""C1 (BASELINE.json configs[0]: 512x512, max_iter 256, fp64, default view) on the GPU, back to back, and the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
from oracle import oracle
r = fr.Renderer(0)
s = torch.cuda.Stream()
W = H = 512
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
st = fr.FractalState(max_iterations=256)
def run(n):
    for _ in range(n):
        r.render(st, W, H, rgba=out, sync=False, stream=s.cuda_stream)
run(20); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s); run(500); e1.record(s); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 500
print("C1 GPU: %.4f ms/frame = %.0f Mpx/s (stages %d)" % (ms, W * H / ms / 1e3, r.last_stages()))
p = oracle.OracleParams(max_iterations=256)
for th in (1, 16):
    oracle.render(p, W, H, threads=th, planes=False)
    t0 = time.perf_counter()
    for _ in range(5): oracle.render(p, W, H, threads=th, planes=False)
    dt = (time.perf_counter() - t0) / 5
    print("C1 oracle, %2d thread(s): %.2f ms/frame = %.1f Mpx/s" % (th, dt * 1e3, W * H / dt / 1e6))
