import sys, statistics
sys.path.insert(0, "/root/repo")
import torch, fractalrenderer_amd as fr
r = fr.Renderer(0); W = H = 4096
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
for name, kw in (("stripes only", dict(stripe_enabled=True)), ("trap", dict(orbit_trap_enabled=True)), ("trap+stripes", dict(orbit_trap_enabled=True, stripe_enabled=True))):
    st = fr.FractalState(max_iterations=1024, **kw)
    ts = []
    for k in range(6):
        r.render(st, W, H, rgba=out)
        if k: ts.append(r.last_kernel_ms())
    print("%-14s %.3f ms" % (name, statistics.median(ts)))
