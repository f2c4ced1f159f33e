#!/usr/bin/env python3
"""Small frames (the default view at the reference's interactive settings: max_iter 256, fp32 and fp64) over sizes, 8 against 64
queue shards against the automatic choice, interleaved.  usage: small_frames.py [rounds] [max_iter]"""
import os, random, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 21
r = fr.Renderer(0)
random.seed(2)
st = fr.FractalState(max_iterations=int(sys.argv[2]) if len(sys.argv) > 2 else 256)
for W, H in ((256, 256), (400, 300), (512, 512), (640, 480), (800, 600), (1024, 768), (1280, 720)):
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    for prec in (fr.Precision.F32, fr.Precision.F64):
        t = {0: [], 8: [], 64: []}
        for rd in range(rounds + 1):
            order = [0, 8, 64]; random.shuffle(order)
            for sh in order:
                r.set_option("shards", sh)
                r.render(st, W, H, precision=prec, rgba=out)
                if rd: t[sh].append(r.last_kernel_ms())
        a, b, c = (statistics.median(t[k]) for k in (0, 8, 64))
        print(f"{W:5d}x{H:<5d} {prec.name}: automatic {a * 1e3:7.1f} us   8 shards {b * 1e3:7.1f}   64 shards {c * 1e3:7.1f}   grid {r.last_grid()}", flush=True)
