"""One problem of tools/soak_r5.py's multi-wave generator (seed on the command line) in detail: where the map gradients of the HIP
route differ from the float64 oracle's.  MF_MW_BWD=0 python tools/debug_soak_mw.py 146  = the general backward on the same problem."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_rollout_gpu import make_dphysics
from tests import helpers as hp
from monoforce_amd import synthetic as syn, _timing
from oracle import dphysics_oracle as orc
DEV = 'cuda'
seed = int(sys.argv[1])
rng = np.random.RandomState(seed)
N = int(rng.choice([5, 7, 8, 9, 16, 17, 32, 33, 50, 64, 65, 100, 128, 129, 175, 223, 256, 257, 300]))
B = int(rng.randint(1, 41)); T = int(rng.choice([2, 3, 5, 17, 40, 80, 120]))
nt = int(rng.choice([2, 4])); integ = int(rng.randint(0, 2)); shared = bool(rng.randint(0, 2)); use_mu = bool(rng.randint(0, 3))
rs_ = float(rng.choice([0.05, 0.1])); d_max = 3.2; xs_only = bool(rng.randint(0, 2))
pts, masks = syn.robot_points_box(N, seed=seed, n_tracks=nt)
nb = 1 if shared else B
z = torch.stack([syn.bump_terrain(syn.bump_params(seed + b, smooth=bool(rng.randint(0, 2))), d_max, rs_, torch.float64) * 0.3 for b in range(nb)]).float()
mu = torch.stack([syn.wave_friction(d_max, rs_, 0.5, 1.0, 1.0 + 0.1 * b, 0.8, torch.float64) for b in range(nb)]).float()
ctrl = syn.varying_controls(B, T, seed=seed, dtype=torch.float64).float()
where = rng.choice(['centre', 'edge', 'off'])
x0 = torch.zeros(B, 3); x0[:, 0] = {'centre': 0.0, 'edge': d_max - 0.3, 'off': d_max + 0.5}[where]; x0[:, 1] = torch.from_numpy(rng.uniform(-1, 1, B)).float()
yaw = torch.from_numpy(rng.uniform(-3.1, 3.1, B)).float()
R0 = torch.zeros(B, 3, 3); R0[:, 0, 0] = yaw.cos(); R0[:, 0, 1] = -yaw.sin(); R0[:, 1, 0] = yaw.sin(); R0[:, 1, 1] = yaw.cos(); R0[:, 2, 2] = 1
xd0 = torch.stack([yaw.cos(), yaw.sin(), torch.zeros(B)], 1) * 0.8
state = (x0, xd0, R0, torch.zeros(B, 3))
print(dict(N=N, B=B, T=T, nt=nt, integ=integ, shared=shared, mu=use_mu, res=rs_, where=str(where), xs=xs_only), flush=True)
ex = lambda m: None if m is None else (m.expand(B, -1, -1) if m.shape[0] == 1 else m)  # noqa: E731
only = [int(v) for v in os.environ.get('DBG_ROLLOUTS', '').split(',') if v]


def grads(fn, dev, dt, sel=None):
    zl, cl = z.clone().to(dt).to(dev).requires_grad_(True), ctrl.clone().to(dt).to(dev).requires_grad_(True)
    ml = mu.clone().to(dt).to(dev).requires_grad_(True) if use_mu else None
    st = [t.clone().to(dt).to(dev) for t in state]
    outs = fn(ex(zl), cl, ex(ml), tuple(st))
    X = outs[0]
    w = syn.probe_weights(outs[0][:, ::3].shape, 0.3, dtype=dt).to(dev)
    if xs_only:
        per = (X[:, ::3] * w).flatten(1).sum(1)
    else:
        per = None
    loss = (X[:, ::3] * w).sum() if xs_only else hp.probe_loss(outs, dt)
    if sel is not None and per is not None:
        loss = per[sel].sum()
    loss.backward()
    return zl.grad.cpu(), (ml.grad.cpu() if use_mu else None), cl.grad.cpu(), [o.detach().cpu() for o in outs]
dp = make_dphysics(pts, masks, integ, rs_, d_max)
spec = hp.spec_from(pts, masks, integ, rs_, d_max)
_timing.start()
gz, gm, gc, outs = grads(lambda zz, cc, mm, st: (lambda so_fo: list(so_fo[0]) + list(so_fo[1]))(dp(zz, cc, state=st, friction=mm)), DEV, torch.float32)
print(_timing.launches()); _timing.stop()
rz, rm, rc, routs = grads(lambda zz, cc, mm, st: (lambda so_fo: list(so_fo[0]) + list(so_fo[1]))(orc.rollout(spec, zz, cc, state=st, friction=mm)), 'cpu', torch.float64)
r32 = grads(lambda zz, cc, mm, st: (lambda so_fo: list(so_fo[0]) + list(so_fo[1]))(orc.rollout(spec, zz, cc, state=st, friction=mm)), 'cpu', torch.float32)
print('rel err gz', hp.rel_err(gz, rz), 'oracle f32', hp.rel_err(r32[0], rz), ' gc', hp.rel_err(gc, rc), 'oracle f32', hp.rel_err(r32[2], rc))
print('outputs: Xs err', hp.rel_err(outs[0], routs[0]), 'oracle f32', hp.rel_err(r32[3][0], routs[0]))
d = (gz.double() - rz).abs()
flat = d.flatten().topk(8)
H = gz.shape[-1]
for v, i in zip(flat.values, flat.indices):
    i = int(i); bb = i // (H * H); ix = (i % (H * H)) // H; iy = i % H
    print('  cell map %d (ix %d, iy %d): hip %.6e  oracle64 %.6e  oracle32 %.6e' % (bb, ix, iy, float(gz.flatten()[i]), float(rz.flatten()[i]), float(r32[0].flatten()[i])))
print('sum of gz: hip %.6e oracle %.6e ; absmax oracle %.3e' % (float(gz.double().sum()), float(rz.sum()), float(rz.abs().max())))
# per-rollout final position (which rollouts sit at the edge)
Xf = routs[0][:, -1]
print('rollouts with |x| or |y| > d_max - 2 cells at the end:', [int(b) for b in range(B) if float(Xf[b, :2].abs().max()) > d_max - 2 * rs_])
if os.environ.get('DBG_EACH'):
    f_h = lambda zz, cc, mm, st: (lambda so_fo: list(so_fo[0]) + list(so_fo[1]))(dp(zz, cc, state=st, friction=mm))
    f_o = lambda zz, cc, mm, st: (lambda so_fo: list(so_fo[0]) + list(so_fo[1]))(orc.rollout(spec, zz, cc, state=st, friction=mm))
    full_ctrl, full_state = ctrl, state
    for b in range(B):
        ctrl = full_ctrl[b:b + 1]; state = tuple(t[b:b + 1] for t in full_state)
        ex = lambda m: None if m is None else m[:1]  # noqa: E731
        B1 = 1
        gz1, _, gc1, o1 = grads(f_h, DEV, torch.float32)
        rz1, _, rc1, ro1 = grads(f_o, 'cpu', torch.float64)
        Tn = int(os.environ.get('DBG_T', '0'))
        print('rollout %2d: gz err %.3e (absmax %.3e)  gc err %.3e  Xs err %.2e  x0 = (%.2f, %.2f) yaw %.2f  max px over time %.2f' % (
            b, hp.rel_err(gz1, rz1), float(rz1.abs().max()), hp.rel_err(gc1, rc1), hp.rel_err(o1[0], ro1[0]), float(full_state[0][b, 0]), float(full_state[0][b, 1]), float(yaw[b]),
            float(ro1[0][0, :, 0].max())), flush=True)
    if os.environ.get('DBG_F64'):
        b = int(os.environ['DBG_F64'])
        ctrl = full_ctrl[b:b + 1]; state = tuple(t[b:b + 1] for t in full_state)
        gz1, _, gc1, o1 = grads(f_h, DEV, torch.float64)
        rz1, _, rc1, ro1 = grads(f_o, 'cpu', torch.float64)
        print('rollout %d in float64 on both sides: gz err %.3e  gc err %.3e  Xs err %.2e' % (b, hp.rel_err(gz1, rz1), hp.rel_err(gc1, rc1), hp.rel_err(o1[0], ro1[0])))
        # float32 HIP forward vs float64 oracle forward, step by step: where do the positions part?
        _, _, _, o32 = grads(f_h, DEV, torch.float32)
        dX = (o32[0].double() - ro1[0]).abs().amax(-1)[0]
        dF = (o32[4].double() - ro1[4]).abs().amax(-1).amax(-1)[0]
        print('|Xs32 - Xs64| over time:', ['%.1e' % float(v) for v in dX[::8]])
        print('|Fs32 - Fs64| over time:', ['%.1e' % float(v) for v in dF[::8]])
