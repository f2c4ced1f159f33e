#!/usr/bin/env python3
"""Lane occupancy of the lane-pool pass: useful lane-updates (from the iter plane) against the lane-update slots the
waves paid for (wave clock x 64, diagnostic build -DFR_STAMP).  usage: FR_LIB_PATH=build/ab/stamp.so pool_occupancy.py workload [opt=v ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
w = WORKLOADS[sys.argv[1]]
W, H = w["W"], w["H"]
r = fr.Renderer(0)
r.set_option("periodicity", -1)
for a in sys.argv[2:]:
    k, v = a.split("="); r.set_option(k, int(v, 0))
rgba = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
it = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]])
state = fr.FractalState(**w["state"])
mi = state.max_iterations
r.render(state, W, H, rgba=rgba, iter=it, **kw)
nw = r.compute_units * 8 * 4
diag = torch.zeros((2, nw, 8), dtype=torch.int64, device="cuda:0")
r.set_option("diag_stride", nw * 8); r.set_option("diag_buffer", diag.data_ptr())
r.render(state, W, H, rgba=rgba, **kw)
ms = r.last_kernel_ms()
d = diag.cpu().numpy()
ran = d[1, :, 1] > 0
slots = float(d[1, ran, 5].sum()) * 64
ex = torch.where(it < mi, it.to(torch.int64) + 1, torch.full_like(it, mi, dtype=torch.int64))
for b0 in (32, 64, 128, 192):
    surv = ex > b0
    useful = float((ex[surv] - b0).sum())
    print(f"b0 {b0:4d}: survivors {int(surv.sum()):9d} ({100*float(surv.float().mean()):.1f} %), useful pool lane-updates {useful:.4g}, "
          f"slots {slots:.4g} -> occupancy {useful/slots:.3f}" if slots else "")
print(f"total executed {float(ex.sum()):.4g}; kernel {ms:.4f} ms; pool waves {int(ran.sum())}, mean wave clock {d[1, ran, 5].mean():.0f}")
