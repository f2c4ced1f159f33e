#!/usr/bin/env python3
"""Back-to-back frame times at interactive sizes (the reference's draw loop): fp32, post chain, default view."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractalrenderer_amd as fr
r = fr.Renderer(0)
s = torch.cuda.Stream()
for W, H, mi in ((1280, 720, 256), (1920, 1080, 256), (1920, 1080, 1024), (3840, 2160, 256), (3840, 2160, 1024)):
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    st = fr.FractalState(max_iterations=mi)
    for prec in (fr.Precision.F32, fr.Precision.F64):
        def run(n):
            for _ in range(n):
                r.render(st, W, H, precision=prec, post_chain=True, rgba=out, sync=False, stream=s.cuda_stream)
        run(10); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); run(200); e1.record(s); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        print("%4dx%-4d max_iter %4d %s: %.4f ms/frame (%.0f fps, %.0f Mpx/s), stages %d" % (W, H, mi, prec.name, ms, 1e3 / ms, W * H / ms / 1e3, r.last_stages()), flush=True)
