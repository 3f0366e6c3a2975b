import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_api_cache_gpu import _problem
from monoforce.losses import physics_loss
dp, z0, mu0, ctrl, states_gt, pred_ts, gt_ts = _problem(seed=1)
z, mu = z0.clone().unsqueeze(0).requires_grad_(True), mu0.clone().unsqueeze(0).requires_grad_(True)
for i in range(9):
    z.grad = mu.grad = None
    states, forces = dp(z_grid=z, controls=ctrl, friction=mu)
    cache = dp._api_step_cache
    loss = physics_loss(states_pred=states, states_gt=states_gt, pred_ts=pred_ts, gt_ts=gt_ts, gamma=0.9)
    if i == 3:
        loss = loss + 1e-3 * (states[1] ** 2).mean()
    pools = dp.__dict__.get('_grad_pools', {})
    info = {k[:3]: (float(p.buf.abs().max()), p.busy, p.pinned) for k, p in pools.items()}
    loss.backward()
    torch.cuda.synchronize()
    e = cache.entry
    sets = e['sets'] if e else []
    print(i, 'replays', cache.replays, 'loss', float(loss.detach()), 'gz', float(z.grad.abs().max()), 'static grads', [float(s.grads[0].abs().max()) for s in sets], 'pools', info,
          'after', {k[:3]: float(p.buf.abs().max()) for k, p in pools.items()}, type(loss.grad_fn).__name__)
