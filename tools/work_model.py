#!/usr/bin/env python3
"""Offline work model of the escape-time kernel from the oracle's iter plane (CPU only).
Prices a frame in wave-iterations for a given sub-tile shape: divergence (lanes idle while the
wave's slowest lane runs), and how many iterations run in tested vs unchecked blocks."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

def model(it, max_iter, fw, fh, K=16):
    H, W = it.shape
    ex = np.where(it < max_iter, it + 1, max_iter).astype(np.int64)
    t = ex.reshape(H // fh, fh, W // fw, fw).transpose(0, 2, 1, 3).reshape(-1, fw * fh)
    wave = t.max(axis=1)
    lane_iters = int(ex.sum()); wave_iters = int(wave.sum())
    # block accounting: block b = iterations [bK, bK+K); a block is "dirty" if any lane escapes in it
    nb = (max_iter + K - 1) // K
    esc = np.where(t < max_iter, (t - 1) // K, -1)          # block index of each escaping lane
    interior = (t >= max_iter) & (np.take_along_axis(t, np.argmax(t, 1)[:, None], 1) >= 0)
    tested = fast = wasted = 0
    for row, wmax in zip(esc, wave):
        dirty = np.zeros(nb + 1, bool); d = row[row >= 0]; dirty[d] = True
        last = (wmax - 1) // K                              # last block the wave runs
        mode_fast = False
        for b in range(last + 1):
            n = min(K, max_iter - b * K)
            if mode_fast and n == K:
                if dirty[b]: wasted += n; tested += n; mode_fast = False
                else: fast += n
            else:
                tested += n; mode_fast = not dirty[b]
    return lane_iters, wave_iters, tested, fast, wasted

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    mi = 1024
    t0 = time.time(); f = O.render(O.OracleParams(max_iterations=mi), N, N); print("oracle %.1fs" % (time.time() - t0))
    for fw, fh in ((8, 8), (16, 4), (64, 1)):
        li, wi, te, fa, wa = model(f.iter, mi, fw, fh)
        cyc = te * 8.5 * 4.25 + (fa + wa) * 6 * 4.25      # ~8.5 VALU-equivalent slots tested, 6 unchecked
        print(f"{fw}x{fh}: lane-iters {li/N/N:.1f}/px  wave-iters*64 {wi*64/N/N:.1f}/px  divergence eff {li/(wi*64):.3f}  "
              f"tested {te*64/N/N:.1f} fast {fa*64/N/N:.1f} wasted {wa*64/N/N:.1f} per px; est cycles/px {cyc*64/N/N/64:.1f}")
