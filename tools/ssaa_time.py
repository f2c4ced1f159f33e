import sys, torch
sys.path.insert(0, "/root/repo")
import fractalrenderer_amd as fr
r = fr.Renderer(0)
if len(sys.argv) > 1:
    r.set_option("periodicity", int(sys.argv[1]))      # usage: ssaa_time.py [periodicity]
    print("periodicity", sys.argv[1])
W = H = 4096
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
for name, ft, kw in (("c2", fr.FractalType.Mandelbrot, dict(max_iterations=1024)),
                     ("c3", fr.FractalType.JuliaSet, dict(max_iterations=2048, center_x=0.0, julia_c_real=-0.8, julia_c_imag=0.156)),
                     ("c5v", fr.FractalType.Mandelbrot, dict(max_iterations=4096, center_x=-0.743643887037151, center_y=0.13182590420533, zoom=0.008))):
    prec = fr.Precision.F32 if name == "c3" else fr.Precision.F64
    for aa in (1, 2, 3):
        st = fr.FractalState(antialiasing_samples=aa, **kw)
        for _ in range(3): r.render(st, W, H, fractal_type=ft, precision=prec, rgba=out)
        ms = r.last_kernel_ms()
        print("%s aa=%d: %.3f ms  (%.3f ms per sample-frame)" % (name, aa, ms, ms / (aa * aa)), flush=True)
