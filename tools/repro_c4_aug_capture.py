"""Repro: EncoderTrainStep(graph=True) with cache_plan=False (bench.py c4_aug) -- print the full traceback of a failing capture."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_problem
from monoforce_amd.terrain_encoder import LiftSplatShoot
from monoforce_amd.train import EncoderTrainStep, synthetic_encoder_batch
dev = torch.device('cuda', 0)
cfg, dp, pts, masks, z, mu, ctrl = build_problem(256, 500, 4, dev, 1)
gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
torch.manual_seed(0)
enc = LiftSplatShoot(gc, dict(final_dim=(256, 512))).to(dev).train()
enc.cache_plan = False
batch = synthetic_encoder_batch(enc, dp, n_rollouts=256, device=dev)
step = EncoderTrainStep(enc, dp, lr=1e-4, graph=True)
os.environ['MF_DEBUG_CAPTURE'] = '1'
try:
    for i in range(3):
        batch[0][4].copy_(batch[0][4] * 1.0)
        print('step', i, float(step.step(batch)[0]), 'graph' if step.graph else 'eager')
except Exception:
    traceback.print_exc()
