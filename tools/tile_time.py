#!/usr/bin/env python3
"""Duration of each launch of a staged render (tile pass, lane-pool pass) from the kernels' per-wave timeline
(diag buffer: first wave start -> last wave end per stage), median over N renders, next to the frame's kernel time
without the diag writes.  The library under test is picked with FR_LIB_PATH (A/B: tools/ab_tile.sh).
usage: tile_time.py [rounds] workload [workload ...]"""
import os, statistics, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 15
# the first renders of a process run on a cold chip (clocks, power state: the first workload listed measured up to 10 %
# slower than the same one listed later): warm it up
_w = WORKLOADS["c2"]; _r = fr.Renderer(0)
_o = torch.empty((_w["H"], _w["W"], 4), dtype=torch.float32, device="cuda:0")
for _ in range(300):
    _r.render(fr.FractalState(**_w["state"]), _w["W"], _w["H"], fractal_type=fr.FractalType[_w["fractal"]], precision=fr.Precision[_w["precision"]], rgba=_o)
torch.cuda.synchronize(); del _o, _r
for name in args or ["c2", "c3"]:
    opts = []; plane = "rgba"
    if ":" in name:
        name, o = name.split(":", 1); opts = o.split(",")
        for kv in list(opts):
            if kv.startswith("plane="): plane = kv.split("=")[1]; opts.remove(kv)
    w = WORKLOADS[name]; W, H = w["W"], w["H"]
    r = fr.Renderer(0)
    r.set_option("periodicity", -1)
    for kv in opts:
        k, v = kv.split("="); r.set_option(k, int(v, 0))
    state = fr.FractalState(**w["state"])
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0") if plane == "rgba" else \
        torch.empty((H, W), dtype=torch.int32, device="cuda:0") if plane == "iter" else \
        torch.empty((H, W), dtype=torch.float64 if w["precision"] == "F64" else torch.float32, device="cuda:0")
    kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]], **{plane: out})
    plain = []
    for i in range(rounds + 3):
        r.render(state, W, H, **kw)
        if i >= 3: plain.append(r.last_kernel_ms())
    nw = r.compute_units * 8 * 4
    nst = r.last_stages()
    diag = torch.zeros((nst, nw, 4), dtype=torch.int64, device="cuda:0")
    r.set_option("diag_stride", nw * 4)
    r.set_option("diag_buffer", diag.data_ptr())
    spans = [[] for _ in range(nst)]; gaps = []
    for i in range(rounds + 2):
        diag.zero_()
        r.render(state, W, H, **kw)
        d = diag.cpu().numpy()
        if i < 2: continue
        ends = []
        for s in range(nst):
            ran = d[s][:, 1] > 0
            lo, hi = d[s][ran, 0].min(), d[s][ran, 1].max()
            spans[s].append((hi - lo) / 100.0); ends.append((lo, hi))
        if nst > 1: gaps.append((ends[1][0] - ends[0][1]) / 100.0)
    txt = "  ".join(f"stage{s} {statistics.median(spans[s]):7.1f} us (min {min(spans[s]):7.1f})" for s in range(nst))
    print(f"{name:8s} {','.join(opts + ([] if plane == 'rgba' else ['plane=' + plane])):24s} frame {statistics.median(plain):.4f} ms (min {min(plain):.4f})  {txt}"
          + (f"  gap {statistics.median(gaps):.1f} us" if gaps else ""), flush=True)
