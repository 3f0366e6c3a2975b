"""torch-profiler view of the terrain encoder's inference forward (4 cameras 3x256x512 -> 256x256 BEV heads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import build_problem
from monoforce_amd.terrain_encoder import LiftSplatShoot
from monoforce_amd.train import synthetic_encoder_batch
dev = torch.device('cuda')
cfg, dp, pts, masks, z, mu, ctrl = build_problem(64, 500, 4, dev, 1)
torch.manual_seed(0)
gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
enc = LiftSplatShoot(gc, dict(final_dim=(256, 512))).to(dev).eval()
batch = synthetic_encoder_batch(enc, dp, n_rollouts=64, device=dev, seed=0)
inputs = batch[0]
with torch.no_grad():
    for _ in range(5): out = enc(*inputs)
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for _ in range(10): out = enc(*inputs)
    torch.cuda.synchronize()
    print('encoder forward ms', (time.perf_counter() - t) / 10 * 1e3)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
        for _ in range(3): out = enc(*inputs)
        torch.cuda.synchronize()
print(p.key_averages().table(sort_by='self_cuda_time_total', row_limit=28, max_name_column_width=90))
