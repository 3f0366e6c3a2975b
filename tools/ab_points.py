"""A/B timing of the rollout kernels for larger bodies (N contact points); MONOFORCE_HIP_LIB selects the library build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing
DEV = 'cuda'
B = int(os.environ.get('AB_B', '1024'))
for N in [int(x) for x in os.environ.get('AB_N', '32,100,175,223').split(',')]:
    for forces in (True, False):
        cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, N, DEV, 1)
        dp.return_forces = forces
        zl, ml = z.to(DEV).clone().requires_grad_(True), mu.to(DEV).clone().requires_grad_(True)
        cd = ctrl.to(DEV)
        def step():
            (Xs, _, _, _), _ = dp(zl.unsqueeze(0), cd, friction=ml.unsqueeze(0))
            (Xs[:, ::10] ** 2).mean().backward()
        step(); step()
        _timing.start()
        for _ in range(4): step()
        k = {n: float(np.mean(v)) for n, v in _timing.stop().items()}
        print(os.environ.get('MONOFORCE_HIP_LIB', 'base')[-38:], 'B', B, 'N', N, 'forces' if forces else 'states', {n: round(v, 3) for n, v in k.items()}, flush=True)
        del dp, zl, ml, cd
