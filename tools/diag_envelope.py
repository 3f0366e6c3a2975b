"""Diagnostic: fp32 free-run error of the HIP rollout vs the reference's fp32 golden, next to the reference's own
fp32-vs-fp64 envelope (T=500, 256x256).  Prints per-rollout curves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers as hp
from tests.test_rollout_gpu import make_dphysics, run_hip

g = hp.load('rollout_full')
pts, masks, z, mu, ctrl = hp.full_inputs(torch.float32)
steps = [10, 50, 100, 150, 200, 250, 300, 350, 400, 450, 499]
for integ in (0, 1):
    dp = make_dphysics(pts, masks, integ, hp.FULL['grid_res'], hp.FULL['d_max'])
    outs, _ = run_hip(dp, z, ctrl, None, mu)
    for k, o in zip(hp.OUT_KEYS[:4], outs[:4]):
        r32, r64 = g[f'f32/i{integ}/{k}'].astype(np.float64), g[f'f64/i{integ}/{k}']
        o = o.numpy().astype(np.float64)
        B, T = r32.shape[:2]
        scale = np.abs(r64).reshape(B, -1).max(1).clip(1e-30)[:, None]
        env = np.abs(r32 - r64).reshape(B, T, -1).max(2) / scale
        err = np.abs(o - r32).reshape(B, T, -1).max(2) / scale
        e64 = np.abs(o - r64).reshape(B, T, -1).max(2) / scale
        for b in range(B):
            print(f'integ={integ} {k:3s} b={b} env  ' + ' '.join(f'{env[b, s]:.1e}' for s in steps))
            print(f'integ={integ} {k:3s} b={b} err  ' + ' '.join(f'{err[b, s]:.1e}' for s in steps))
            print(f'integ={integ} {k:3s} b={b} e64  ' + ' '.join(f'{e64[b, s]:.1e}' for s in steps))
