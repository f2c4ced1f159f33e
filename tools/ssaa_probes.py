import sys, statistics, random
sys.path.insert(0, "/root/repo")
import torch, fractalrenderer_amd as fr
r = fr.Renderer(0)
cases = [("hd aa2 mi256 f32", 1920, 1080, fr.Precision.F32, dict(max_iterations=256, antialiasing_samples=2)),
         ("hd aa2 mi1024 f64", 1920, 1080, fr.Precision.F64, dict(max_iterations=1024, antialiasing_samples=2)),
         ("720p aa3 mi256 f32", 1280, 720, fr.Precision.F32, dict(max_iterations=256, antialiasing_samples=3)),
         ("4096 c5v aa2", 4096, 4096, fr.Precision.F64, dict(max_iterations=4096, antialiasing_samples=2, center_x=-0.743643887037151, center_y=0.13182590420533, zoom=0.008)),
         ("4096 c2 aa2", 4096, 4096, fr.Precision.F64, dict(max_iterations=1024, antialiasing_samples=2)),
         ("hd trap mi256 f32", 1920, 1080, fr.Precision.F32, dict(max_iterations=256, orbit_trap_enabled=True)),
         ("4096 trap mi1024 f64", 4096, 4096, fr.Precision.F64, dict(max_iterations=1024, orbit_trap_enabled=True))]
for name, W, H, prec, kw in cases:
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    st = fr.FractalState(**kw)
    t = {p: [] for p in (8, 4, 2, 1)}
    for rd in range(7):
        order = list(t); random.shuffle(order)
        for p in order:
            r.set_option("probes", p)
            r.render(st, W, H, precision=prec, rgba=out)
            if rd: t[p].append(r.last_kernel_ms())
    print("%-22s " % name + "  ".join("probes=%d %.4f" % (p, statistics.median(v)) for p, v in t.items()), flush=True)
