#!/usr/bin/env python3
"""Debug: dump the state of the first lane-pool wave whose stretch loop trips the watchdog (build with -DFR_WATCHDOG_DUMP)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import fractalrenderer_amd as fr
r = fr.Renderer(0)
st = fr.FractalState(center_x=-0.743643887037151, center_y=0.13182590420533, zoom=1e-6, max_iterations=16384)
W, H = 40, 24
nu = torch.empty((H, W), dtype=torch.float64, device="cuda:0")
diag = torch.zeros(4096, dtype=torch.int64, device="cuda:0")
for window in (1, 16, 48, 4096):
    r.set_option("periodicity", window); r.set_option("staging", 3)
    r.set_option("diag_stride", 2048); r.set_option("diag_buffer", diag.data_ptr() + 0)
    try:
        for k in range(3):
            diag.zero_(); torch.cuda.synchronize()
            r.set_option("diag_buffer", diag.data_ptr() + 2048 * 8 * 0)
            r.render(st, W, H, nu=nu)
        print("window", window, "ok")
    except fr.FractalRendererError as e:
        d = diag.cpu().numpy()
        # the pool pass is stage 1: its diag region starts at diag_stride words
        base = 2048
        names = ["wclock", "next_deadline", "have_running", "goal", "newly", "nactive", "stride", "fast", "watchdog", "snap_window", "next_snap", "dry", "life_avg", "refill_at", "mi|i0"]
        print("window", window, "TRIPPED:", {n: int(d[base + i]) for i, n in enumerate(names)})
        lanes = d[base + 16: base + 80]; fl = d[base + 80: base + 144]
        for l in range(64):
            print(f"  lane {l:2d} pixel {int(lanes[l] >> 32) & 0xFFFFFFFF:10d} deadline-wclock {int(np.int32(lanes[l] & 0xFFFFFFFF)):8d} fin {int(fl[l] >> 32)} cyc {int(fl[l] & 0xFFFFFFFF)}")
        break
