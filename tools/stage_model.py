#!/usr/bin/env python3
"""CPU model (oracle iter plane) of multi-stage stream compaction: stage 0 runs every 8x8 sub-tile up
to S0 iterations; pixels still alive are compacted (tile order) into dense 64-lane waves for the
next stage, and so on.  Reports wave-iterations (x64) per pixel per stage vs the single-pass kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

def tile_order(a, fw=8, fh=8):
    H, W = a.shape
    return a.reshape(H // fh, fh, W // fw, fw).transpose(0, 2, 1, 3).reshape(-1)

def model(ex, max_iter, bounds):
    """ex: executed iterations per pixel in tile order (1..max_iter)."""
    n = ex.size
    res = []
    alive = np.arange(n)
    lo = 0
    for hi in bounds:
        e = ex[alive]
        pad = (-len(e)) % 64
        ee = np.concatenate([e, np.zeros(pad, e.dtype)]).reshape(-1, 64)
        wave = np.clip(ee, lo, hi).max(axis=1) - lo            # iterations the wave runs in this stage
        wave = np.ceil(wave / 16) * 16 if hi != max_iter else wave
        lane = (np.clip(e, lo, hi) - lo).sum()
        res.append((lo, hi, len(e), wave.sum() * 64, lane))
        alive = alive[ex[alive] > hi] if hi < max_iter else alive[:0]
        lo = hi
    return res

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    cases = {"c2": (O.OracleParams(max_iterations=1024), 1024),
             "c3": (O.OracleParams(fractal=1, precision=0, center_x=0.0, zoom=3.0, max_iterations=2048,
                                   julia_c_real=-0.8, julia_c_imag=0.156), 2048)}
    for name, (p, mi) in cases.items():
        f = O.render(p, N, N)
        ex = tile_order(np.where(f.iter < mi, f.iter + 1, mi).astype(np.int64))
        single = ex.reshape(-1, 64).max(axis=1).sum() * 64
        print(f"{name} {N}^2: lane-iters/px {ex.mean():.1f}; single pass wave-iters*64/px {single/ex.size:.1f} (eff {ex.sum()/single:.3f})")
        for bounds in ([32, mi], [32, 128, mi], [32, 128, 512, mi], [16, 64, 256, mi], [64, 256, mi], [32, 64, 128, 256, 512, mi], [48, 192, 768, mi]):
            bounds = sorted(set(b for b in bounds if b <= mi))
            r = model(ex, mi, bounds)
            tot = sum(x[3] for x in r)
            surv = sum(x[2] for x in r[1:])
            print(f"   {str(bounds):34s} wave-iters*64/px {tot/ex.size:7.1f} (eff {ex.sum()/tot:.3f}); survivor records written {surv/ex.size:.3f}/px; "
                  + " ".join(f"[{lo}-{hi}: {cnt/ex.size:.3f}px {w/ex.size:.1f}]" for lo, hi, cnt, w, l in r))
