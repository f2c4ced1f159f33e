#!/usr/bin/env python3
"""Renders one workload a few times (for rocprofv3 passes).
usage: run_case.py workload [max_iter] [reps] [plane] [nparts]   (nparts > 1: part 0 of a row-strip sharded frame)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
w = WORKLOADS[sys.argv[1]]
st = dict(w["state"])
if len(sys.argv) > 2 and int(sys.argv[2]) > 0: st["max_iterations"] = int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
plane = sys.argv[4] if len(sys.argv) > 4 else "rgba"
W, H = w["W"], w["H"]
nparts = int(sys.argv[5]) if len(sys.argv) > 5 else 1
from fractalrenderer_amd.distributed import pick_rows_per_strip
shard = fr.Shard(0, nparts, pick_rows_per_strip(H, nparts)) if nparts > 1 else None
rows = shard.rows(H) if shard else H
r = fr.Renderer(0)
prec = fr.Precision[w["precision"]]
bufs = {"rgba": torch.empty((rows, W, 4), dtype=torch.float32, device="cuda:0"),
        "nu": torch.empty((rows, W), dtype=torch.float64 if w["precision"] == "F64" else torch.float32, device="cuda:0"),
        "iter": torch.empty((rows, W), dtype=torch.int32, device="cuda:0")}
for _ in range(reps):
    r.render(fr.FractalState(**st), W, H, fractal_type=fr.FractalType[w["fractal"]], precision=prec, shard=shard, **{plane: bufs[plane]})
print("kernel ms", r.last_kernel_ms())
