#!/usr/bin/env python3
"""Randomised soak of fr_node on ONE card (every part on device 0): random part counts, layouts, strip heights, roots, views,
fractals, precisions, plane sets, device / host planes, and (round 4) frame slots, render lanes and bursts of 1-4 frames in flight
waited for in a random order -- every frame bitwise against fr_render.  usage: soak_node.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fractalrenderer_amd as fr
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
r = fr.Renderer(0)
nodes = {}
bad = 0
for t in range(trials):
    n = int(rng.choice([1, 2, 3, 4, 5, 8, 13, 16]))
    node = nodes.get(n) or nodes.setdefault(n, fr.Node([0] * n))
    ft = fr.FractalType([0, 1, 2, 5][int(rng.integers(0, 4))])
    prec = fr.Precision.F32 if ft == fr.FractalType.Deep_Zoom or rng.random() < 0.4 else fr.Precision.F64
    W, H = int(rng.integers(9, 700)), int(rng.integers(5, 500))
    zoom = float(10.0 ** rng.uniform(-3.0, 0.5))
    kw = dict(max_iterations=int(rng.choice([40, 129, 300, 1024, 2500])), center_x=-0.74 + zoom * float(rng.uniform(-0.3, 0.3)),
              center_y=0.13 + zoom * float(rng.uniform(-0.3, 0.3)), zoom=zoom, palette_mode=int(rng.integers(0, 4)),
              antialiasing_samples=int(rng.choice([1, 1, 1, 2])))
    if ft == fr.FractalType.Mandelbrot and rng.random() < 0.35:       # the shader's effects: the lean kernels' code-3 instantiations
        kw["orbit_trap_enabled"] = bool(rng.integers(0, 2))
        kw["stripe_enabled"] = bool(rng.integers(0, 2)) or not kw["orbit_trap_enabled"]
        kw["interior_style"] = int(rng.choice([0, 0, 1, 2]))
    if ft == fr.FractalType.Deep_Zoom:
        kw["use_perturbation"] = bool(rng.integers(0, 2)); kw["antialiasing_samples"] = 1
    nu_dt = torch.float64 if prec == fr.Precision.F64 else torch.float32
    post = bool(rng.integers(0, 2)) and ft != fr.FractalType.Deep_Zoom
    node.set_option("layout", int(rng.integers(0, 2)))
    node.set_option("rows_per_strip", int(rng.choice([0, 0, 1, 5, 8, 24, 32, 100])))
    node.set_option("periodicity", int(rng.choice([0, -1, 1, 48])))
    node.set_option("slots", int(rng.choice([1, 2, 2, 3, 8])))
    node.set_option("lanes", int(rng.choice([1, 2, 2, 3])))
    planes = [bool(rng.integers(0, 2)) for _ in range(3)]
    if not any(planes): planes[0] = True
    host = rng.random() < 0.3
    burst = int(rng.choice([1, 1, 2, 3, 4]))            # frames in flight: the same geometry, views a little apart
    frames = []
    for b in range(burst):
        kb = dict(kw, center_x=kw["center_x"] + 0.07 * b * zoom, max_iterations=kw["max_iterations"] + 16 * b)
        st = fr.FractalState(**kb)
        want = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda"), torch.empty((H, W), dtype=nu_dt, device="cuda"),
                torch.empty((H, W), dtype=torch.int32, device="cuda")]
        r.render(st, W, H, fractal_type=ft, precision=prec, post_chain=post, rgba=want[0], nu=want[1], iter=want[2])
        got = [(np.zeros(tuple(w.shape), dtype=w.cpu().numpy().dtype) if host else torch.zeros_like(w)) if use else None for w, use in zip(want, planes)]
        frames.append((st, want, got, int(rng.integers(0, n))))
    torch.cuda.synchronize()                              # the fills of `got` run on torch's stream
    if burst == 1 and rng.random() < 0.5:
        st, want, got, root = frames[0]
        node.render(st, W, H, root=root, fractal_type=ft, precision=prec, post_chain=post, rgba=got[0], nu=got[1], iter=got[2],
                    sync=bool(rng.integers(0, 2)))
        node.wait()
    else:
        tickets = [node.submit(st, W, H, root=root, fractal_type=ft, precision=prec, post_chain=post, rgba=got[0], nu=got[1], iter=got[2])
                   for st, want, got, root in frames]
        for i in rng.permutation(burst):
            node.wait_frame(tickets[int(i)])
        assert node.in_flight() == 0
    for st, want, got, root in frames:
        ok = True
        for w, g in zip(want, got):
            if g is None: continue
            same = np.array_equal(w.cpu().numpy(), g) if host else bool(torch.equal(w, g))
            if not same:
                ok = False
                break
        if not ok:
            bad += 1
            print("MISMATCH trial", t, n, ft, prec, W, H, kw, planes, host, root, burst)
            break
    if (t + 1) % 25 == 0: print("trial", t + 1, "failures", bad, flush=True)
print("done: %d trials, %d failures" % (trials, bad))
sys.exit(1 if bad else 0)
