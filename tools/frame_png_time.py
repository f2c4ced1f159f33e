#!/usr/bin/env python3
"""End-to-end time of the RenderFrameCallback body (render -> 8-bit export -> PNG) per frame size."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fractalrenderer_amd as fr
r = fr.Renderer(0)
st = fr.FractalState(max_iterations=1024)
d = tempfile.mkdtemp()
for W, H in ((1920, 1080), (3840, 2160), (4096, 4096), (8192, 8192)):
    for threads in ("1", ""):
        if threads: os.environ["FR_PNG_THREADS"] = threads
        else: os.environ.pop("FR_PNG_THREADS", None)
        p = os.path.join(d, "f.png")
        r.render_frame(st, W, H, p)
        t0 = time.perf_counter()
        n = 2
        for _ in range(n): assert r.render_frame(st, W, H, p)
        dt = (time.perf_counter() - t0) / n
        print("%5dx%-5d threads %-4s %.3f s/frame  (%.1f Mpx/s, file %.1f MB)" % (W, H, threads or "all", dt, W * H / dt / 1e6, os.path.getsize(p) / 1e6), flush=True)
print("nproc", os.cpu_count())
