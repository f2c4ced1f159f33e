import time, torch, sys
sys.path.insert(0, "/root/repo")
import fractalrenderer_amd as fr
r = fr.Renderer(0)
dev = torch.device("cuda:0")
st = fr.FractalState(max_iterations=1024)
W, H = 64, 64
nu = torch.empty((H // 8, W), dtype=torch.float64, device=dev)
s = torch.cuda.Stream()
sh = fr.Shard(0, 8, 8)
for _ in range(50):
    r.render(st, W, H, nu=nu, shard=sh, sync=False, stream=s.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 2000
for _ in range(n):
    r.render(st, W, H, nu=nu, shard=sh, sync=False, stream=s.cuda_stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host per async render call: %.1f us (enqueue), drained after %.1f us/call" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
p = st.to_params()
t0 = time.perf_counter()
for _ in range(n):
    st.to_params()
print("to_params: %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
a = torch.empty((64, 8, 32, 512), device=dev); b = torch.empty((64, 32, 512), device=dev)
t0 = time.perf_counter()
for _ in range(n):
    a[:, 3].copy_(b)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("strided copy_ host: %.1f us" % ((t1 - t0) / n * 1e6))
ev = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(n):
    with torch.cuda.stream(s):
        ev.record(s)
    s.wait_event(ev)
print("stream ctx + event record + wait: %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
