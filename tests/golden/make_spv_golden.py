"""Generates tests/golden/spv_frames.npz by executing the reference's compiled shaders.

    python tests/golden/make_spv_golden.py [case ...]

Reads /root/reference/FractalRenderer/shaders/<shader>.comp.spv at generation time (the binaries are
never copied), runs one interpreter invocation per pixel (tests/golden/spirv_interp.py) and stores:
  <case>/rgba    float32 (H, W, 4)  the texel the invocation wrote with OpImageWrite
  <case>/iter    int32   (H, W)     escape index of the sample function (aa == 1 cases only)
  <case>/smooth  float32 (H, W)     its smooth value where one was computed, NaN elsewhere (aa == 1)
  __meta__       JSON: sha256 of every shader binary that was executed, numpy version
Push constants are the 20 floats oracle.pack_push_constants produces (pinned against the reference's
packing by tests/test_host.py); the Deep_Zoom orbit buffer is the fp64 orbit narrowed to float pairs, as
DeepZoomManager uploads it (src/deep_zoom_system.cpp:102-110).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, ".."), os.path.join(HERE, "..", "..")]

from spirv_interp import F32, Cell, Invocation, Module   # noqa: E402
from spv_cases import SPV_CASES                          # noqa: E402
from oracle import oracle                                # noqa: E402

SHADERS = "/root/reference/FractalRenderer/shaders"
PROBES = {   # shader -> (escape index, smooth value) as "function:variable"
    "mandelbrot": ("compute_sample:i", "compute_sample:nu"),
    "julia": ("sample_julia:iter", "sample_julia:smooth_val"),
    "burning_ship": ("compute_burning_ship:iter", "compute_burning_ship:smooth_val"),
    "test_deep_zoom": ("main:i", "get_color:smooth_iter"),
}


def run_case(shader, p, W, H):
    m = Module(os.path.join(SHADERS, shader + ".comp.spv"))
    pc = oracle.pack_push_constants(p)
    gid = m.global_named("gl_GlobalInvocationID")
    push = next(g for g, s in m.global_storage.items() if s == 9)
    image = next(g for g, s in m.global_storage.items() if s == 0)
    buffers = [g for g, s in m.global_storage.items() if s in (2, 12)]
    orbit = None
    if buffers:
        if p.use_perturbation:
            orbit = [[F32(x), F32(y)] for x, y in oracle.reference_orbit(p.center_x, p.center_y, p.max_iterations)]
        else:
            orbit = []
        assert int(pc[13]) == len(orbit)
    rgba = np.zeros((H, W, 4), np.float32)
    it = np.full((H, W), -1, np.int32)
    sm = np.full((H, W), np.nan, np.float32)
    probes = PROBES[shader]
    steps = 0
    for y in range(H):
        for x in range(W):
            g = {gid: Cell([x, y, 0]), image: Cell(None),
                 push: Cell([[F32(v) for v in pc[4 * k:4 * k + 4]] for k in range(len(pc) // 4)])}
            for b in buffers:
                g[b] = Cell([orbit])
            inv = Invocation(m, g, (W, H), probe=probes).run()
            assert len(inv.stores) == 1 and inv.stores[0][1] == [x, y]
            rgba[y, x] = inv.stores[0][2]
            it[y, x] = inv.probes.get(probes[0], -1)
            sm[y, x] = inv.probes.get(probes[1], np.nan)
            steps += inv.steps
    return rgba, it, sm, steps


def main(argv):
    names = argv or list(SPV_CASES)
    path = os.path.join(HERE, "spv_frames.npz")
    out = dict(np.load(path)) if argv and os.path.exists(path) else {}
    for name in names:
        shader, p, W, H = SPV_CASES[name]
        t = time.time()
        rgba, it, sm, steps = run_case(shader, p, W, H)
        out[name + "/rgba"] = rgba
        if p.aa == 1:
            out[name + "/iter"] = it
            out[name + "/smooth"] = sm
        print("%-28s %-14s %3dx%-3d %9d SPIR-V instructions  %.1f s" % (name, shader, W, H, steps, time.time() - t), flush=True)
    meta = json.loads(str(out["__meta__"])) if "__meta__" in out else {"sha256": {}}
    for shader in sorted({SPV_CASES[n][0] for n in names}):
        with open(os.path.join(SHADERS, shader + ".comp.spv"), "rb") as f:
            meta["sha256"][shader + ".comp.spv"] = hashlib.sha256(f.read()).hexdigest()
    meta["numpy"] = np.__version__
    out["__meta__"] = np.array(json.dumps(meta, sort_keys=True))
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1:])
