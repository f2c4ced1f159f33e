"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/fr_oracle.c).

The reference has no golden vectors and cannot run here, so these
are outputs of the build's own restatement: they guard the oracle against regressions and
give the GPU box a checker that does not depend on compiling anything there.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from cases import CASES, MANDEL_PALETTES, JULIA_PALETTES  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    O.build()
    frames = {}
    for name, (p, W, H) in CASES.items():
        f = O.render(p, W, H, threads=1)
        frames[name + "/iter"] = f.iter
        frames[name + "/nu"] = f.nu
        frames[name + "/rgba"] = f.rgba
        frames[name + "/executed"] = np.int64(f.executed)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **frames)

    ts = np.linspace(-0.25, 1.25, 385, dtype=np.float32)
    pal = {"t": ts}
    for m in MANDEL_PALETTES:
        pal["mandelbrot/%d" % m] = np.stack([O.palette(0, m, float(t)) for t in ts])
    for m in JULIA_PALETTES:
        pal["julia/%d" % m] = np.stack([O.palette(1, m, float(t)) for t in ts])
    np.savez_compressed(os.path.join(HERE, "palettes.npz"), **pal)
    sz = sum(os.path.getsize(os.path.join(HERE, f)) for f in ("frames.npz", "palettes.npz"))
    print("wrote frames.npz + palettes.npz, %d bytes" % sz)


if __name__ == "__main__":
    main()
