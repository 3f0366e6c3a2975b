"""Quick device check: gradients of the component-parallel backward vs the one-point-per-lane backward over short horizons
(T = 1 .. 48; a bug in the loop structure shows at T = 2, one in the unroll parity at T = 3 / 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_rollout_gpu import make_dphysics
from tests import helpers as hp
from monoforce_amd import synthetic as syn
pts, masks = syn.robot_points_4()
B = 6
z = torch.stack([syn.bump_terrain(syn.bump_params(20 + k), 1.6, 0.1) * 0.3 for k in range(B)])
mu = torch.stack([syn.wave_friction(1.6, 0.1, 0.5, 1.0, 1.1 + 0.2 * k, 0.8) for k in range(B)])
INTEG = int(os.environ.get('MF_INTEG', '1'))      # 1: the default integrator, 0: dynamics()
for T in (1, 2, 3, 4, 5, 6, 7, 8, 20, 47, 48):
    ctrl = syn.varying_controls(B, max(T, 2), seed=3)[:, :T]
    res = {}
    for ppl in (16, 1):
        dp = make_dphysics(pts, masks, INTEG, 0.1, 1.6, points_per_lane=ppl)
        zd, md, cd = z.cuda().requires_grad_(True), mu.cuda().requires_grad_(True), ctrl.cuda().requires_grad_(True)
        st, fo = dp(zd, cd, friction=md)
        hp.probe_loss(list(st) + list(fo), torch.float32).backward()
        res[ppl] = (zd.grad.cpu(), md.grad.cpu(), cd.grad.cpu())
    print('integ', INTEG, 'T', T, ' '.join(f'{n}={hp.rel_err(a, b):.2e}' for n, a, b in zip(('gz', 'gmu', 'gctrl'), res[16], res[1])), flush=True)
