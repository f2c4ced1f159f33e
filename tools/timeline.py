#!/usr/bin/env python3
"""Per-wave timeline of one render from the kernels' diag buffer, per stage: when waves start/finish,
how the work and the dequeues are spread.
usage: timeline.py workload [max_iter] [opt=value ...] [nparts=N (part 0 of a sharded frame)] [plane=nu|rgba]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
w = WORKLOADS[sys.argv[1]]
st = dict(w["state"])
args = sys.argv[2:]
if args and args[0].isdigit():
    st["max_iterations"] = int(args[0]); args = args[1:]
W, H = w["W"], w["H"]
r = fr.Renderer(0)
nparts, plane = 1, "rgba"
for a in list(args):
    k, v = a.split("=")
    if k == "nparts": nparts = int(v); args.remove(a)
    elif k == "plane": plane = v; args.remove(a)
    else: r.set_option(k, int(v, 0))
from fractalrenderer_amd.distributed import pick_rows_per_strip
shard = fr.Shard(0, nparts, pick_rows_per_strip(H, nparts)) if nparts > 1 else None
rows = shard.rows(H) if shard else H
out = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda:0") if plane == "rgba" else \
    torch.empty((rows, W), dtype=torch.int32, device="cuda:0") if plane == "iter" else \
    torch.empty((rows, W), dtype=torch.float64 if w["precision"] == "F64" else torch.float32, device="cuda:0")
kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]], shard=shard, **{plane: out})
state = fr.FractalState(**st)
for _ in range(3):
    r.render(state, W, H, **kw)
base_ms = r.last_kernel_ms()
nw = r.compute_units * 8 * 4
nst = r.last_stages()
diag = torch.zeros((nst, nw, 4), dtype=torch.int64, device="cuda:0")
r.set_option("diag_stride", nw * 4)
r.set_option("diag_buffer", diag.data_ptr())
for _ in range(3):
    r.render(state, W, H, **kw)
ms = r.last_kernel_ms()
d = diag.cpu().numpy()
t0 = d[d[:, :, 1] > 0][:, 0].min()
print(f"{sys.argv[1]} {st} {' '.join(args)}: kernel {base_ms:.4f} ms ({ms:.4f} with diag), {nst} stage(s), grid {r.last_grid()} WG")
for s in range(nst):
    ds = d[s]; ran = ds[:, 1] > 0
    if not ran.any():
        print(f"  stage {s}: no waves"); continue
    start = (ds[ran, 0] - t0) / 100.0; end = (ds[ran, 1] - t0) / 100.0
    items, claims = ds[ran, 2], ds[ran, 3] & 0xFFFFFFFF
    dry = (ds[ran, 3] >> 32) / 100.0 + start          # lane-pool passes: when the wave found the queue dry
    lo, hi = start.min(), end.max()
    edges = np.linspace(lo, hi, 11)
    live = [(np.minimum(end, b) - np.maximum(start, a)).clip(min=0).sum() / max(b - a, 1e-9) for a, b in zip(edges[:-1], edges[1:])]
    worked = items > 0
    if worked.sum() > 8 and np.ptp(items[worked]) > 0:
        dur = (end - start)[worked]
        A = np.vstack([np.ones(worked.sum()), items[worked], claims[worked]]).T
        coef = np.linalg.lstsq(A, dur, rcond=None)[0]
        print(f"           wave time ~ {coef[0]:.1f} us + {coef[1]:.2f} us/item + {coef[2]:.2f} us/dequeue  (start spread {np.percentile(start,99)-lo:.1f} us)")
    if (ds[ran, 3] >> 32).any():
        after = end - dry
        print(f"           queue dry seen at p1 {np.percentile(dry,1)-lo:.1f} p50 {np.percentile(dry,50)-lo:.1f} p99 {np.percentile(dry,99)-lo:.1f} us; "
              f"run-out after dry: p50 {np.percentile(after,50):.1f} p90 {np.percentile(after,90):.1f} p99 {np.percentile(after,99):.1f} max {after.max():.1f} us")
    print(f"  stage {s}: {lo:8.1f} -> {hi:8.1f} us ({hi-lo:7.1f}), waves {ran.sum()} ({worked.sum()} with work), items/wave mean {items[worked].mean() if worked.any() else 0:.1f} "
          f"max {items.max()}, dequeues {claims.sum()}, finish p50 {np.percentile(end,50)-lo:.1f} p99 {np.percentile(end,99)-lo:.1f}; live/10%: " + " ".join(f"{x:.0f}" for x in live))
