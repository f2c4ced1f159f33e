#!/usr/bin/env python3
"""One frame as K row bands on K render contexts / streams (bands are shards with rows_per_strip = H/K, so their
packed outputs are the bands of the full frame): do the bands' phase boundaries overlap each other's bodies?
usage: split_frame.py [workload ...]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
from bench import WORKLOADS
names = sys.argv[1:] or ["c2", "c3", "c5"]
ctxs = [fr.Renderer(0) for _ in range(4)]
for kv in filter(None, os.environ.get("FR_SPLIT_OPTS", "").split(",")):      # e.g. FR_SPLIT_OPTS=periodicity=-1
    for c in ctxs: c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
STRIPS = int(os.environ.get("FR_SPLIT_STRIPS", "0"))                           # rows per strip: interleaved strips instead of bands
streams = [torch.cuda.Stream() for _ in range(4)]
for name in names:
    w = WORKLOADS[name]; W, H = w["W"], w["H"]
    st = fr.FractalState(**w["state"])
    kw = dict(fractal_type=fr.FractalType[w["fractal"]], precision=fr.Precision[w["precision"]])
    ref = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    ctxs[0].render(st, W, H, rgba=ref, **kw)
    main = torch.cuda.current_stream()
    outs = {K: torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for K in (1, 2, 4)}

    def frame(K):
        out, R = outs[K], H // K
        fork = torch.cuda.Event(); fork.record(main)
        joins = []
        for k in range(K):
            s = streams[k]
            s.wait_event(fork)
            if STRIPS and K > 1:             # whole-frame planes are not exposed to Python: packed part buffers (not compared)
                sh = fr.Shard(k, K, STRIPS)
                ctxs[k].render(st, W, H, rgba=out[k * R:k * R + sh.rows(H)], shard=sh, sync=False, stream=s.cuda_stream, **kw)
            else:
                ctxs[k].render(st, W, H, rgba=out[k * R:(k + 1) * R], shard=fr.Shard(k, K, R) if K > 1 else None,
                               sync=False, stream=s.cuda_stream, **kw)
            e = torch.cuda.Event(); e.record(s); joins.append(e)
        for e in joins: main.wait_event(e)

    times = {K: [] for K in outs}
    for rd in range(13):
        for K in outs:                      # interleaved: 20 frames back to back per measurement
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for _ in range(20): frame(K)
            e1.record(main); torch.cuda.synchronize()
            if rd: times[K].append(e0.elapsed_time(e1) / 20)
    for K, t in times.items():
        print("%-3s %d band(s): median %.4f ms min %.4f  identical %s" % (name, K, statistics.median(t), min(t), torch.equal(outs[K], ref)), flush=True)
