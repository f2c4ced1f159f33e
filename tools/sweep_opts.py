#!/usr/bin/env python3
"""Interleaved A/B of named option sets.  usage: sweep_opts.py workload rounds "k=v,k=v" "k=v" ... ("" = defaults)"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
ALL = ("workgroups_per_cu", "run_max", "run_min", "shift_bias", "subtile_shape", "staging", "stage_first",
       "stream_run_max", "stream_run_min", "stream_workgroups_per_cu", "pool_refill_at", 
       "probes", "stream_probes", "stream_rotate", "periodicity", "tile_kernel", "tile_pixels", "tile_exit", "tile_exit_from", "prepare", "stripes", "shards", "regions")
name, rounds = sys.argv[1], int(sys.argv[2])
variants = sys.argv[3:] or [""]
w = WORKLOADS[name]; W, H = w["W"], w["H"]
state = fr.FractalState(**w["state"])
kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]],
          rgba=torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0"))
r = fr.Renderer(0)
times = {v: [] for v in variants}
import random
random.seed(1)
for rd in range(rounds + 1):
    order = list(variants)
    random.shuffle(order)          # neighbours influence each other through clocks and power: shuffle per round
    for v in order:
        for k in ALL: r.set_option(k, 0)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("="); r.set_option(k, int(val, 0))
        r.render(state, W, H, **kw)
        if rd: times[v].append(r.last_kernel_ms())
print(f"workload {name} {W}x{H} rounds {rounds}: options -> median ms, min ms, Mpx/s")
for v, t in sorted(times.items(), key=lambda kv: statistics.median(kv[1])):
    med = statistics.median(t); print(f"  {v or '(defaults)':60s} {med:.4f} {min(t):.4f} {W*H/med/1e3:.0f}")
