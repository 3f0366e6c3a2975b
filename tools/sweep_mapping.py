"""Time both lane mappings of the rollout kernels over B (forward and backward), to place the crossover."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing
DEV = 'cuda'
for B in (4096, 8192, 16384, 32768, 65536):
    for ppl in (1, 4):
        cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, 4, DEV, 1)
        dp.points_per_lane = ppl
        zl, ml = z.to(DEV).clone().requires_grad_(True), mu.to(DEV).clone().requires_grad_(True)
        cd = ctrl.to(DEV)
        def step():
            (Xs, _, _, _), _ = dp(zl.unsqueeze(0), cd, friction=ml.unsqueeze(0))
            (Xs[:, ::10] ** 2).mean().backward()
        step(); step()
        _timing.start()
        for _ in range(4): step()
        k = {n: float(np.mean(v)) for n, v in _timing.stop().items()}
        print(B, ppl, {n: round(v, 3) for n, v in k.items()}, 'fwd %.1f%% bwd %.1f%%' % (304*B*500/k['rollout_fwd_kernel']/1e6/80, 640*B*500/k['rollout_bwd_kernel']/1e6/80))
        del dp, zl, ml, cd
