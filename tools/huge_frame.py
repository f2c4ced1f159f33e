import sys, torch, time
sys.path.insert(0, "/root/repo")
import fractalrenderer_amd as fr
r = fr.Renderer(0)
W = H = 16384
st = fr.FractalState(max_iterations=1024)
it = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
rgba = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
r.render(st, W, H, rgba=rgba, iter=it)
t0 = time.perf_counter(); r.render(st, W, H, rgba=rgba, iter=it); dt = time.perf_counter() - t0
print("16384^2: %.2f ms host-timed, kernel %.2f ms, %.0f Mpx/s" % (dt * 1e3, r.last_kernel_ms(), W * H / r.last_kernel_ms() / 1e3))
print("interior fraction", float((it == 1024).float().mean()))
# conjugate symmetry about the row y = H/2 (pixel rows y and H - y map to +-Im): rows 1..H-1
top = it[1:H // 2]; bot = torch.flip(it[H // 2 + 1:], dims=[0])
print("conjugate-symmetric rows identical:", bool(torch.equal(top, bot)))
print("alpha all one:", bool((rgba[..., 3] == 1).all()), "free MB", torch.cuda.mem_get_info()[0] >> 20)
