#!/usr/bin/env python3
"""Throughput of whole C2 frames with 1, 2 and 3 frames in flight (one render context + stream each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
W = H = 4096
st = fr.FractalState(max_iterations=1024)
dev = torch.device("cuda:0")
for k in (1, 2, 3, 1, 2):
    rs = [fr.Renderer(0) for _ in range(k)]
    ss = [torch.cuda.Stream(device=dev) for _ in range(k)]
    bufs = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(k)]
    def run(n):
        for i in range(n):
            rs[i % k].render(st, W, H, rgba=bufs[i % k], sync=False, stream=ss[i % k].cuda_stream)
    run(6); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    for s in ss: s.wait_stream(main)
    e0.record(main)
    for s in ss: s.wait_event(e0)
    n = 60
    run(n)
    for s in ss: main.wait_stream(s)
    e1.record(main)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%d frame(s) in flight: %.4f ms/frame = %.0f Mpx/s" % (k, ms, W * H / ms / 1e3), flush=True)
    for r in rs: r.close()
