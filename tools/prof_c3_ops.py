import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import build_problem
from monoforce_amd.train import TerrainFitProblem
dev = torch.device('cuda')
cfg, dp, pts, masks, z, mu, ctrl = build_problem(1024, 500, 4, dev, 1)
zt, mt = z.to(dev), mu.to(dev)
prob = TerrainFitProblem(dp, zt, mt, ctrl.to(dev))
zl = (zt * 0.9).clone().requires_grad_(True); ml = mt.clone().requires_grad_(True)
for _ in range(3): prob.step(zl, ml)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
    for _ in range(5): prob.step(zl, ml)
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by='cuda_time_total', row_limit=30, max_name_column_width=70))
ev = [e for e in p.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print(len(ev) / 5, 'device kernels per step')
