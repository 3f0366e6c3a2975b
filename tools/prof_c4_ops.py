"""torch-profiler view of the encoder train step (bench workload c4): where the step's GPU time goes, steady state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import build_problem
from monoforce_amd.terrain_encoder import LiftSplatShoot
from monoforce_amd.train import EncoderTrainStep, synthetic_encoder_batch
dev = torch.device('cuda')
cfg, dp, pts, masks, z, mu, ctrl = build_problem(1024, 500, 4, dev, 1)
torch.manual_seed(0)
gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
enc = LiftSplatShoot(gc, dict(final_dim=(256, 512))).to(dev).train()
ebatch = synthetic_encoder_batch(enc, dp, n_rollouts=1024, device=dev, seed=0)
estep = EncoderTrainStep(enc, dp, lr=1e-4)
for _ in range(6): estep.step(ebatch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
    for _ in range(3): estep.step(ebatch)
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=90))
