"""Backward of large bodies vs the number of private gradient copies of a shared map (dphysics_bwd.GRAD_COPIES): AB_B, AB_N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing, dphysics_bwd
B = int(os.environ.get('AB_B', '256')); N = int(os.environ.get('AB_N', '223'))
for copies in (16, 64, 256):
    dphysics_bwd.grad_copies_for = lambda B_, N_, c=copies: max(1, min(c, B_))
    cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, N, 'cuda', 1)
    dp.return_forces = False
    zl, ml = z.cuda().clone().requires_grad_(True), mu.cuda().clone().requires_grad_(True)
    cd = ctrl.cuda()
    def step():
        (Xs, _, _, _), _ = dp(zl.unsqueeze(0), cd, friction=ml.unsqueeze(0))
        (Xs[:, ::10] ** 2).mean().backward()
    step(); step()
    _timing.start()
    for _ in range(4): step()
    k = {n: round(float(np.mean(v)), 3) for n, v in _timing.stop().items()}
    print('B', B, 'N', N, 'GRAD_COPIES', copies, k, flush=True)
