"""One problem of tools/soak_win.py (seed on the command line) in detail: error of every gradient against the float64 oracle, per selected
rollout (through its control gradient), the same under the float64 HIP build.  MF_BWD_WIN=0 python tools/debug_soak_win.py 53 = the atomics route."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_rollout_gpu import make_dphysics
from tests import helpers as hp
from monoforce_amd import synthetic as syn, _timing
from oracle import dphysics_oracle as orc
DEV = 'cuda'
seed = int(sys.argv[1])
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
assert not scattered, 'default starts only here'
print(dict(B=B, T=T, H=H, N=N, integ=integ, mu=friction), 'WIN', os.environ.get('MF_BWD_WIN', '1'), flush=True)
spec = hp.spec_from(pts, masks, integ, res, 6.4)
wts = syn.probe_weights((sel.numel(), T, 3), phase=0.1 * seed)


def hip(dt, Bn=B, rows=None):
    dp = make_dphysics(pts, masks, integ, res, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    c = ctrl if rows is None else ctrl[rows]
    leaf = lambda t: t.to(dt).to(DEV).detach().clone().requires_grad_(True)  # noqa: E731
    zd = leaf(z); md = leaf(mu) if friction else None; cd = leaf(c)
    _timing.start()
    (Xs, _, _, _), _ = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0) if friction else None)
    s = sel if rows is None else torch.arange(len(rows))
    (Xs[s.to(DEV)] * wts.to(dt).to(DEV)).sum().backward()
    nm = _timing.launches().get('rollout_bwd_kernel', '?').split(' grid')[0][-70:]
    _timing.stop()
    return zd.grad.cpu(), cd.grad[s.to(DEV)].cpu(), (md.grad.cpu() if friction else None), Xs[s.to(DEV)].detach().cpu(), nm


def oracle(dt):
    zc = z.to(dt).requires_grad_(True); mc = mu.to(dt).requires_grad_(True) if friction else None; cc = ctrl[sel].to(dt).requires_grad_(True)
    m = sel.numel()
    (rX, _, _, _), _ = orc.rollout(spec, zc.unsqueeze(0).expand(m, -1, -1), cc, friction=mc.unsqueeze(0).expand(m, -1, -1) if friction else None)
    (rX * wts.to(dt)).sum().backward()
    return zc.grad, cc.grad, (mc.grad if friction else None), rX.detach()
r64, r32 = oracle(torch.float64), oracle(torch.float32)
g = hip(torch.float32)
print('kernel', g[4])
for nm, i in (('z', 0), ('controls', 1), ('mu', 2)):
    if g[i] is None: continue
    print('%-9s hip32 vs oracle64 %.3e   oracle32 vs oracle64 %.3e' % (nm, hp.rel_err(g[i], r64[i]), hp.rel_err(r32[i], r64[i])))
print('Xs        hip32 vs oracle64 %.3e   oracle32 vs oracle64 %.3e' % (hp.rel_err(g[3], r64[3]), hp.rel_err(r32[3], r64[3])))
sc = float(r64[1].abs().max())
for k in range(sel.numel()):
    e = float((g[1][k].double() - r64[1][k]).abs().max()) / sc
    eo = float((r32[1][k].double() - r64[1][k]).abs().max()) / sc
    ex = float((g[3][k].double() - r64[3][k]).abs().max())
    if e > 1e-4 or eo > 1e-4:
        print('  rollout %6d: controls-gradient error %.3e (oracle32 %.3e)  |dXs| %.2e  v %.3f w %.3f' % (int(sel[k]), e, eo, ex, float(ctrl[sel[k], 0, 0]), float(ctrl[sel[k], 0, 1])))
d = (g[0].double() - r64[0]).abs()
top = d.flatten().topk(6)
for v, i in zip(top.values, top.indices):
    i = int(i); print('  cell (ix %d, iy %d): hip %.5e oracle64 %.5e oracle32 %.5e' % (i // H, i % H, float(g[0].flatten()[i]), float(r64[0].flatten()[i]), float(r32[0].flatten()[i])))
print('sum gz: hip %.6e oracle64 %.6e' % (float(g[0].double().sum()), float(r64[0].sum())))
if os.environ.get('DBG_SUBSET'):      # only the selected rollouts as their own small batch (other kernels), float32 and float64
    for dt in (torch.float32, torch.float64):
        gs_ = hip(dt, rows=sel)
        print('selected rollouts alone, %s: kernel %s  z %.3e controls %.3e' % (str(dt)[6:], gs_[4], hp.rel_err(gs_[0], r64[0]), hp.rel_err(gs_[1], r64[1])))
