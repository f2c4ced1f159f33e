#!/usr/bin/env python3
"""Where a wave's time goes (diagnostic build with -DFR_STAMP: tools/build_variant.sh work stamp with EXTRA_HIPFLAGS=-DFR_STAMP).
usage: FR_LIB_PATH=build/ab/stamp.so tools/stamps.py workload [opt=value ...]"""
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
for a in sys.argv[2:]:
    k, v = a.split("="); r.set_option(k, int(v, 0))
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]], rgba=out)
state = fr.FractalState(**w["state"])
for _ in range(5):
    r.render(state, W, H, **kw)
nw = r.compute_units * 8 * 4
nst = r.last_stages()
diag = torch.zeros((nst, nw, 8), dtype=torch.int64, device="cuda:0")
r.set_option("diag_stride", nw * 8)
r.set_option("diag_buffer", diag.data_ptr())
for _ in range(3):
    r.render(state, W, H, **kw)
ms = r.last_kernel_ms()
d = diag.cpu().numpy()
GHZ = float(os.environ.get("FR_CLOCK_GHZ", "2.1"))
print(f"{sys.argv[1]}: kernel {ms:.4f} ms with stamps; cycles -> us at {GHZ} GHz")
names = ["dequeue (q.next)", "stream block claim", "refill incl. dequeue + record loads", "retire: shade + store"]
for s in range(nst):
    ds = d[s]; ran = ds[:, 1] > 0
    if not ran.any():
        continue
    life = (ds[ran, 1] - ds[ran, 0]) / 100.0
    print(f"  stage {s}: {ran.sum()} waves, mean wave lifetime {life.mean():.1f} us, items/wave {ds[ran,2].mean():.1f}, claims/wave {(ds[ran,3] & 0xFFFFFFFF).mean():.1f}")
    if s >= 1:
        wc = ds[ran, 5].astype(np.float64)
        print(f"      wave clock (updates run per wave): mean {wc.mean():.0f}, total lane-update slots {wc.sum()*64:.4g}")
    for k in range(4):
        if s >= 1 and k == 1:
            continue
        us = ds[ran, 4 + k] / (GHZ * 1e3)
        print(f"      {names[k]:40s} mean {us.mean():8.2f} us/wave = {100*us.mean()/life.mean():5.1f} % of lifetime (p90 {np.percentile(us,90):.1f})")
