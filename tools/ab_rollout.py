"""A/B timing of the rollout kernels (forward, optionally backward) over B and workgroup size; pick the library build with
MONOFORCE_HIP_LIB=<path to libmonoforce_hip.so>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing
DEV = 'cuda'
Bs = [int(x) for x in os.environ.get('AB_B', '4096,16384,65536').split(',')]
blocks = [int(x) for x in os.environ.get('AB_BLOCK', '64,256').split(',')]
bwd = os.environ.get('AB_BWD', '0') == '1'
for B in Bs:
    for block in blocks:
        for forces in (True, False):
            cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, 4, DEV, 1)
            dp.block = block
            dp.return_forces = forces
            zl, ml = z.to(DEV).clone().requires_grad_(bwd), mu.to(DEV).clone().requires_grad_(bwd)
            cd = ctrl.to(DEV)
            def step():
                (Xs, _, _, _), _ = dp(zl.unsqueeze(0), cd, friction=ml.unsqueeze(0))
                if bwd: (Xs[:, ::10] ** 2).mean().backward()
            step(); step()
            _timing.start()
            for _ in range(4): step()
            k = {n: float(np.mean(v)) for n, v in _timing.stop().items()}
            print(os.environ.get('MONOFORCE_HIP_LIB', 'base')[-40:], B, block, 'forces' if forces else 'states', {n: round(v, 3) for n, v in k.items()}, flush=True)
            del dp, zl, ml, cd
