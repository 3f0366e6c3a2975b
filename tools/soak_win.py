"""Soak of the LDS-window backward kernels (round 5) with the float64 ORACLE as the only referee: random large batches of the <= 4-point
body on one shared map pair -- component-parallel early recompute (4097 .. 8192 rollouts), positions-only one point per lane (beyond) --
over map sizes 64 .. 512 cells (a 512-cell map leaves most of a trajectory OUTSIDE the 128-cell window: the atomics route), start poses
given (scattered over the map) or default, both integrators, with / without a friction map, ragged batches.  The loss touches 24
rollouts spread over the batch (the last ones included); a problem passes when every gradient is within max(2e-4, 3 x the distance
between the oracle's own float32 and float64 gradients).      python tools/soak_win.py [n]      (test infrastructure, as tests/)"""
import os, sys
os.environ.setdefault('OMP_NUM_THREADS', '8')      # the CPU oracle is the referee here: 8 threads run it 5 x faster than 128 (tests/conftest.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(min(int(os.environ['OMP_NUM_THREADS']), torch.get_num_threads()))
from tests.test_rollout_gpu import make_dphysics
from tests import helpers as hp
from monoforce_amd import synthetic as syn, _timing
from oracle import dphysics_oracle as orc
DEV = 'cuda'
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
res_all, worst = [], 0.0
for seed in range(int(os.environ.get('SOAK_SEED0', '0')), int(os.environ.get('SOAK_SEED0', '0')) + n):      # (SOAK_SEED0: another range of problems)
    rng = np.random.RandomState(9000 + seed)
    B = int(rng.choice([4608, 6144, 8192, 8192 + 512, 12288, 16384, 16384 + 37, 24576, 32768, 32768 + 4]))
    T = int(rng.randint(20, 121)); H = int(rng.choice([64, 128, 256, 512])); res = 12.8 / H
    N = int(rng.choice([3, 4])); integ = int(rng.randint(0, 2)); friction = bool(rng.randint(0, 4)); scattered = bool(rng.randint(0, 2))
    pts4, _ = syn.robot_points_4()
    pts = pts4[:N].copy(); masks = [pts[:, 1] > 0, pts[:, 1] <= 0]
    z = syn.bump_terrain(syn.bump_params(seed + 3), 6.4, res) * float(rng.choice([0.3, 1.0]))
    mu = syn.wave_friction(6.4, res) if friction else None
    ctrl = syn.const_controls(B, T, seed=seed)
    sub = 24
    sel = torch.cat([torch.arange(0, B, B // (sub - 3))[:sub - 3], torch.arange(B - 3, B)])
    state = None
    if scattered:
        g = torch.Generator().manual_seed(seed)
        x0 = torch.zeros(B, 3); x0[:, :2] = (torch.rand(B, 2, generator=g) - 0.5) * 12.4
        yaw = torch.rand(B, generator=g) * 6.2831853
        R0 = torch.zeros(B, 3, 3); R0[:, 0, 0] = yaw.cos(); R0[:, 0, 1] = -yaw.sin(); R0[:, 1, 0] = yaw.sin(); R0[:, 1, 1] = yaw.cos(); R0[:, 2, 2] = 1.0
        xd0 = torch.zeros(B, 3); xd0[:, 0] = ctrl[:, 0, 0] * yaw.cos(); xd0[:, 1] = ctrl[:, 0, 0] * yaw.sin()
        w0 = torch.zeros(B, 3); w0[:, 2] = ctrl[:, 0, 1]
        state = (x0, xd0, R0, w0)
    spec = hp.spec_from(pts, masks, integ, res, 6.4)
    wts = syn.probe_weights((sel.numel(), T, 3), phase=0.1 * seed)
    dp = make_dphysics(pts, masks, integ, res, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    zd = z.to(DEV).requires_grad_(True); md = mu.to(DEV).requires_grad_(True) if friction else None; cd = ctrl.to(DEV).requires_grad_(True)
    _timing.start()
    (Xs, _, _, _), _ = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0) if friction else None, state=tuple(t.clone().to(DEV) for t in state) if scattered else None)
    (Xs[sel.to(DEV)] * wts.to(DEV)).sum().backward()
    name = _timing.launches()['rollout_bwd_kernel'].split(' grid')[0]
    _timing.stop()

    def oracle(dt):
        zc = z.to(dt).requires_grad_(True); mc = mu.to(dt).requires_grad_(True) if friction else None; cc = ctrl[sel].to(dt).requires_grad_(True)
        st = tuple(t[sel].clone().to(dt) for t in state) if scattered else None
        m = sel.numel()
        (rX, _, _, _), _ = orc.rollout(spec, zc.unsqueeze(0).expand(m, -1, -1), cc, state=st, friction=mc.unsqueeze(0).expand(m, -1, -1) if friction else None)
        (rX * wts.to(dt)).sum().backward()
        return [zc.grad, cc.grad] + ([mc.grad] if friction else [])
    r64, r32 = oracle(torch.float64), oracle(torch.float32)
    got = [zd.grad.cpu(), cd.grad[sel.to(DEV)].cpu()] + ([md.grad.cpu()] if friction else [])
    ratios = []
    for a, b64, b32 in zip(got, r64, r32):
        bar = max(2e-4, 3.0 * hp.rel_err(b32, b64))
        ratios.append(hp.rel_err(a, b64) / bar)
    rest = torch.ones(B, dtype=torch.bool); rest[sel] = False
    clean = float(cd.grad[rest.to(DEV)].abs().max()) == 0.0
    ok = max(ratios) <= 1.0 and clean and all(bool(torch.isfinite(a).all()) for a in got)
    worst = max(worst, max(ratios))
    res_all.append(ok)
    print(('ok  ' if ok else 'FAIL'), seed, dict(B=B, T=T, H=H, N=N, integ=integ, mu=friction, scattered=scattered), 'worst error / bar %.3f' % max(ratios), name[-60:], flush=True)
    del dp, zd, md, cd, Xs
    torch.cuda.empty_cache()
print(f'LDS-window backward kernels vs the float64 oracle: {len(res_all)} random problems, {sum(res_all)} within the bar; worst error / bar {worst:.2f}')
