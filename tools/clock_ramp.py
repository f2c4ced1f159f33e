import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, fractalrenderer_amd as fr
from bench import WORKLOADS
w = WORKLOADS["c2"]; W, H = w["W"], w["H"]
st = fr.FractalState(**w["state"])
r = fr.Renderer(0, timing=False); r.set_option("periodicity", -1)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
s = torch.cuda.Stream(); h = s.cuda_stream
torch.cuda.synchronize()
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 0.0)
N = 120
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record(s)
for i in range(N):
    r.render(st, W, H, rgba=out, sync=False, stream=h)
    ev[i + 1].record(s)
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print("frames  0-4  :", " ".join("%.3f" % x for x in t[:5]))
print("frames  5-24 : mean %.4f" % (sum(t[5:25]) / 20))
print("frames 25-49 : mean %.4f" % (sum(t[25:50]) / 25))
print("frames 50-119: mean %.4f" % (sum(t[50:]) / 70))
