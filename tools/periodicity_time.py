#!/usr/bin/env python3
"""Lane-pool pass with and without cycle closing ("periodicity" option): identical planes, frame times.
usage: periodicity_time.py [workload ...] [window=N]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS

names = [a for a in sys.argv[1:] if "=" not in a] or ["c2", "c3", "c5", "c4"]
windows = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("window=")] or [1]
r = fr.Renderer(0)
for name in names:
    w = WORKLOADS[name]
    W, H = w["W"], w["H"]
    st = fr.FractalState(**w["state"])
    ft, pr = fr.FractalType[w["fractal"]], fr.Precision[w["precision"]]
    nu_dt = torch.float64 if pr == fr.Precision.F64 else torch.float32
    planes = {}
    for mode in [-1] + windows:        # -1: off (the reference's iteration count)
        try:
            r.set_option("periodicity", mode)
        except fr.FractalRendererError:      # a library from before the option's default became "on": 0 was "off"
            r.set_option("periodicity", 0)
        rgba = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
        nu = torch.empty((H, W), dtype=nu_dt, device="cuda:0")
        it = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
        r.render(st, W, H, fractal_type=ft, precision=pr, rgba=rgba, nu=nu, iter=it)
        ts = []
        for _ in range(3 if name == "c4" else 10):
            r.render(st, W, H, fractal_type=ft, precision=pr, rgba=rgba)
            ts.append(r.last_kernel_ms())
        same = ""
        if mode == -1:
            planes = dict(rgba=rgba, nu=nu, it=it)
        else:
            same = "identical planes: %s" % all(torch.equal(a, b) for a, b in ((rgba, planes["rgba"]), (nu, planes["nu"]), (it, planes["it"])))
        print("%-3s periodicity %-4d  median %.4f ms  min %.4f ms  %8.0f Mpx/s  %s" % (
            name, mode, statistics.median(ts), min(ts), W * H / statistics.median(ts) / 1e3, same), flush=True)
        del rgba, nu, it
r.set_option("periodicity", 0)   # back to the default (on)
