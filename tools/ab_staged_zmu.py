"""Does the component-parallel forward gain from a STAGED interleaved (z, mu) pair at the BASELINE batch?  (c4's forward runs the ZMU
instantiation, c3's the plain one.)  Forward with the record + backward of a positions-only loss, B = AB_B rollouts, with the pair
stage_terrain leaves on z and with plain copies of the same maps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing
from monoforce_amd.terrain_stage import stage_terrain
DEV = 'cuda'
for B in [int(x) for x in os.environ.get('AB_B', '1024,4096').split(',')]:
    cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, 4, DEV, 1)
    dp.return_forces = False
    cd = ctrl.to(DEV)
    geom = z.to(DEV).unsqueeze(0).clone().requires_grad_(True)
    fr = mu.to(DEV).unsqueeze(0).clone()
    for staged in (True, False):
        def step():
            _, zs, ms = stage_terrain(geom, torch.zeros_like(geom), fr, k=1)
            if not staged:
                zs, ms = zs * 1.0, ms * 1.0          # the same values in tensors that carry no staged pair
            (Xs, _, _, _), _ = dp(zs, cd, friction=ms)
            (Xs[:, ::10] ** 2).mean().backward()
        step(); step()
        _timing.start()
        for _ in range(6): step()
        k = {n: round(float(np.mean(v)), 4) for n, v in _timing.stop().items() if 'rollout' in n}
        print('B', B, 'staged (z, mu) pair' if staged else 'plain maps        ', k, flush=True)
