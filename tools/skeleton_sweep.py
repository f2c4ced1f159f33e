#!/usr/bin/env python3
"""Per-pixel skeleton of the tile pass: far-exterior view (every pixel escapes at i <= 1), option sets interleaved.
usage: skeleton_sweep.py plane rounds "k=v,k=v" ..."""
import os, random, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
plane, rounds = sys.argv[1], int(sys.argv[2])
variants = sys.argv[3:] or [""]
W = H = 4096
r = fr.Renderer(0)
bufs = {"rgba": torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0"),
        "nu": torch.empty((H, W), dtype=torch.float64, device="cuda:0"),
        "iter": torch.empty((H, W), dtype=torch.int32, device="cuda:0")}
st = fr.FractalState(center_x=8.0, center_y=8.0, zoom=2.0, max_iterations=1024)
times = {v: [] for v in variants}
used = set()
random.seed(2)
for rd in range(rounds + 1):
    order = list(variants); random.shuffle(order)
    for v in order:
        for k in used: r.set_option(k, 0)
        prec = fr.Precision.F64
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=")
            if k == "f32": prec = fr.Precision.F32; continue
            r.set_option(k, int(val, 0)); used.add(k)
        r.render(st, W, H, precision=prec, **{plane: bufs[plane]} if plane != "nu32" else {"nu": bufs["nu"]})
        if rd: times[v].append(r.last_kernel_ms())
for v, t in sorted(times.items(), key=lambda kv: statistics.median(kv[1])):
    print(f"  {v or '(defaults)':50s} {statistics.median(t):.4f} {min(t):.4f}")
