#!/usr/bin/env python3
"""In-process A/B sweep of launch geometry (interleaved rounds, median + min of the library's own
HIP-event kernel time).  usage: tools/sweep.py [workload] [rounds] [wg,run_max,shape,run_min,shift_bias ...]"""
import itertools
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import fractalrenderer_amd as fr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
w = WORKLOADS[name]
W, H = w["W"], w["H"]
state = fr.FractalState(**w["state"])
ftype, prec = fr.FractalType[w["fractal"]], fr.Precision[w["precision"]]
r = fr.Renderer(0)
rgba = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
variants = [(wg, rmax, 3, rmin, bias) for wg in (4, 8) for rmax in (16, 32, 64) for rmin in (1, 2, 4) for bias in (-2, 0, 2, 4)]
variants += [(8, 32, s, 1, 0) for s in (4, 6)] + [(0, 0, 0, 0, 0)]
if len(sys.argv) > 3:
    variants = [tuple(int(x) for x in v.split(",")) for v in sys.argv[3:]]
times = {v: [] for v in variants}
for rd in range(rounds + 1):
    for v in variants:
        r.set_tuning(v[0], v[1], v[2], v[3], v[4])
        r.render(state, W, H, fractal_type=ftype, precision=prec, rgba=rgba)
        if rd:
            times[v].append(r.last_kernel_ms())
print(f"workload {name}: {W}x{H}, rounds {rounds}; (wg/CU, run_max, shape, run_min, shift_bias) -> median ms, min ms, Mpx/s(median)")
for v, t in sorted(times.items(), key=lambda kv: statistics.median(kv[1])):
    med = statistics.median(t)
    print(f"  {v}: {med:.4f} {min(t):.4f}  {W*H/med/1e3:.0f}")
