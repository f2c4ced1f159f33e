import sys, statistics
sys.path.insert(0, "/root/repo")
import torch, fractalrenderer_amd as fr
r = fr.Renderer(0)
for name, W, H, prec, kw in (("1080p default mi512 f32", 1920, 1080, fr.Precision.F32, dict(max_iterations=512)),
                             ("1080p reset-view mi512 f64", 1920, 1080, fr.Precision.F64, dict(max_iterations=512, zoom=1.5)),
                             ("4K default mi256 f32", 3840, 2160, fr.Precision.F32, dict(max_iterations=256)),
                             ("720p cardioid mi700 f64", 1280, 720, fr.Precision.F64, dict(max_iterations=700, center_x=-0.2, zoom=0.8))):
    st = fr.FractalState(**kw)
    outs = {}
    for per in (-1, 0):      # off / the default (on)
        r.set_option("periodicity", per)
        out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
        ts = []
        for k in range(12):
            r.render(st, W, H, precision=prec, rgba=out)
            if k: ts.append(r.last_kernel_ms())
        outs[per] = (out, statistics.median(ts), r.last_stages())
    print("%-28s stages %d: %.4f ms -> %.4f ms with periodicity, identical %s" % (name, outs[-1][2], outs[-1][1], outs[0][1], torch.equal(outs[-1][0], outs[0][0])), flush=True)
