"""A/B of the lane mappings of the rollout kernels: component-parallel (16 lanes per rollout) vs one point per lane (G = 4),
forward (all outputs / states only) and backward, over the batch size.  AB_B=256,1024,... AB_BWD=1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing
DEV = 'cuda'
Bs = [int(x) for x in os.environ.get('AB_B', '256,1024,2048,4096,8192,16384').split(',')]
bwd = os.environ.get('AB_BWD', '0') == '1'
integ = int(os.environ.get('AB_INTEG', '1'))
only_cp = os.environ.get('AB_ONLY_CP', '0') == '1'      # the component-parallel states-only rows alone (mode sweeps)
private = os.environ.get('AB_PRIVATE', '0') == '1'      # one height / friction map PER ROLLOUT ([B,H,W] in HBM) instead of a shared pair
for B in Bs:
    for ppl in ((16,) if only_cp else (16, 1)):
        for forces in ((False,) if only_cp else (True, False)):
            cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, 4, DEV, integ)
            dp.points_per_lane = ppl
            dp.return_forces = forces
            zl, ml = z.to(DEV).clone(), mu.to(DEV).clone()
            if private:
                zl, ml = zl.unsqueeze(0).repeat(B, 1, 1), ml.unsqueeze(0).repeat(B, 1, 1)
            else:
                zl, ml = zl.unsqueeze(0), ml.unsqueeze(0)
            zl.requires_grad_(bwd); ml.requires_grad_(bwd)
            cd = ctrl.to(DEV)
            def step():
                if bwd: zl.grad = ml.grad = None
                (Xs, _, _, _), _ = dp(zl, cd, friction=ml)
                if bwd: (Xs[:, ::10] ** 2).mean().backward()
            step(); step()
            _timing.start()
            for _ in range(6): step()
            k = {n: float(np.mean(v)) for n, v in _timing.stop().items()}
            print('B', B, 'private maps' if private else 'shared map', 'lanes', 'cp16' if ppl == 16 else 'g4', 'forces' if forces else 'states', {n: round(v, 4) for n, v in k.items()}, flush=True)
            del dp, zl, ml, cd
