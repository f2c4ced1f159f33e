#!/usr/bin/env python3
"""Lane pool: what share of a wave's updates ran tested, in dirty unchecked stretches, and in deferred replays
(diagnostic build: EXTRA_HIPFLAGS="-DFR_STAMP -DFR_STAMP_TESTED" tools/build_variant.sh work stampt).
usage: FR_LIB_PATH=build/ab/stampt.so tools/pool_shares.py workload [opt=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
for name in [a for a in sys.argv[1:] if "=" not in a]:
    w = WORKLOADS[name]
    W, H = w["W"], w["H"]
    r = fr.Renderer(0)
    r.set_option("periodicity", -1)
    for a in sys.argv[1:]:
        if "=" in a:
            k, v = a.split("="); r.set_option(k, int(v, 0))
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]], rgba=out)
    state = fr.FractalState(**w["state"])
    for _ in range(3):
        r.render(state, W, H, **kw)
    nw = r.compute_units * 8 * 4
    nst = r.last_stages()
    diag = torch.zeros((nst, nw, 8), dtype=torch.int64, device="cuda:0")
    r.set_option("diag_stride", nw * 8)
    r.set_option("diag_buffer", diag.data_ptr())
    r.render(state, W, H, **kw)
    d = diag.cpu().numpy()[1]
    ran = d[:, 1] > 0
    tested, clock, dirty, replay = (d[ran, 4 + k].astype(np.float64).sum() for k in range(4))
    print(f"{name}: lane pool, {int(ran.sum())} waves: updates per wave (wave clock) {clock / ran.sum():.0f}; tested {100 * tested / clock:.1f} %, "
          f"in dirty unchecked stretches {100 * dirty / clock:.1f} %, deferred replays {100 * replay / clock:.1f} % of the wave clock "
          f"(replays run beside it: +{100 * replay / clock:.1f} % updates at full occupancy)")
    r.close()
