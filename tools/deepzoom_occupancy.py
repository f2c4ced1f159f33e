import sys, os
sys.path.insert(0, os.getcwd())
import torch, fractalrenderer_amd as fr
from bench import WORKLOADS
w = WORKLOADS["deepzoom"]; W, H = w["W"], w["H"]
st = fr.FractalState(**w["state"])
it = torch.empty((H, W), dtype=torch.int32, device="cuda")
with fr.Renderer(0) as r:
    r.render(st, W, H, fractal_type=fr.FractalType.Deep_Zoom, precision=fr.Precision.F32, iter=it)
mi = st.max_iterations
work = torch.where(it < mi, it + 1, torch.full_like(it, mi)).to(torch.float64)
t = work.view(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
print("mean iter/px", work.mean().item(), "lockstep occupancy (sum / 64 max per 8x8 tile):", (t.sum() / (64 * t.max(dim=1).values.sum())).item(),
      "tiles all-interior:", (t.min(dim=1).values >= mi).float().mean().item())
