"""Round-5 soak with the float64 ORACLE as the only referee: random problems for
  cp   the component-parallel kernels (<= 4 points; tests/test_random_shapes_gpu.py::_cp_case: 1..130 rollouts, 1..69 steps, both
       integrators, all input variants) and
  mw   the recording forward + record-reading backward of 5..300-point bodies (the generator of tools/soak_mw.py),
each against oracle/dphysics_oracle.py in float64 on the same float32-valued inputs.  A problem passes when every gradient is within
max(2e-3, 3 x the distance between the oracle's OWN float32 and float64 gradients of that problem) -- the bar of the test suite.
    python tools/soak_r5.py [n_cp] [n_mw]        (test infrastructure: the oracle is the checker here, as in tests/)"""
import os, sys
os.environ.setdefault('OMP_NUM_THREADS', '8')      # the CPU oracle is the referee here: 8 threads run it 5 x faster than 128 (tests/conftest.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(min(int(os.environ['OMP_NUM_THREADS']), torch.get_num_threads()))
from tests.test_random_shapes_gpu import _cp_case
from tests.test_rollout_gpu import make_dphysics
from tests import helpers as hp
from monoforce_amd import synthetic as syn
from oracle import dphysics_oracle as orc
DEV = 'cuda'
n_cp = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_mw = int(sys.argv[2]) if len(sys.argv) > 2 else 200
BAR = 2e-3
SEED0 = int(os.environ.get('SOAK_SEED0', '0'))      # another range of problems


def judge(tag, seed, info, got, ref64, ref32):
    ratios, raw = [], []
    for a, b64, b32 in zip(got, ref64, ref32):
        bar = max(BAR, 3.0 * hp.rel_err(b32, b64))
        e = hp.rel_err(a, b64)
        ratios.append(e / bar); raw.append(e)
    fin = all(bool(torch.isfinite(a).all()) for a in got)
    ok = fin and max(ratios) <= 1.0
    if not ok:
        print('FAIL', tag, seed, info, ['%.2e' % v for v in raw], 'ratio to bar %.2f' % max(ratios), flush=True)
    return ok, max(raw), max(ratios)


def summary(tag, res):
    oks = sum(r[0] for r in res)
    print(f'{tag}: {len(res)} random problems, {oks} within the bar; worst gradient error {max(r[1] for r in res):.2e} '
          f'(relative to the largest entry), worst error / bar {max(r[2] for r in res):.2f}', flush=True)


# ---- cp -------------------------------------------------------------------------------------------------------------------------
pts4, _ = syn.robot_points_4()
res = []
for seed in range(SEED0, SEED0 + n_cp):
    info, pts, masks, z, mu, ctrl, state, d_max = _cp_case(seed)
    B = info['B']
    base = make_dphysics(pts4, [pts4[:, 1] > 0, pts4[:, 1] <= 0], info['integ'], info['res'], d_max)
    ex = lambda m: None if m is None else (m.expand(B, -1, -1) if info['shared'] and B > 1 else m)  # noqa: E731

    def loss_of(outs, dt):
        if info['loss'] == 0:
            return hp.probe_loss(outs, dt)
        l = (outs[0] * syn.probe_weights(outs[0].shape, phase=0.4).to(outs[0])).sum()
        if info['loss'] == 2:
            l = l + 1e-3 * (outs[4] * syn.probe_weights(outs[4].shape, phase=1.4).to(outs[4])).sum()
        return l
    dp = make_dphysics(pts4, [pts4[:, 1] > 0, pts4[:, 1] <= 0], info['integ'], info['res'], d_max, points_per_lane=16)
    dp.dphys_cfg.robot_points = torch.as_tensor(pts)
    dp.dphys_cfg.driving_parts = [torch.as_tensor(m) for m in masks]
    dp.x_points = dp.dphys_cfg.robot_points.unsqueeze(0).to(dp.device)
    dp._cache = {('iinv', torch.float32): base._iinv(torch.float32)}
    zl = z.clone().to(DEV).requires_grad_(True); cl = ctrl.clone().to(DEV).requires_grad_(True)
    ml = None if mu is None else mu.clone().to(DEV).requires_grad_(True)
    st = None if state is None else [s.clone().to(DEV) for s in state]
    so, fo = dp(ex(zl), cl, state=None if st is None else tuple(st), friction=ex(ml))
    loss_of(list(so) + list(fo), torch.float32).backward()
    got = [zl.grad.cpu(), cl.grad.cpu()] + ([ml.grad.cpu()] if ml is not None else [])
    spec = hp.spec_from(pts, masks, info['integ'], info['res'], d_max)
    spec.robot_size_y = float(pts4[:, 1].max() - pts4[:, 1].min())      # (as the inertia: the HIP side keeps the 4-point body's robot_size; a 1-point body's own y-extent is 0)
    _pi = orc.point_inertia          # (the soak runs the N-point body with the 4-point body's inertia: same for the oracle)
    P4 = torch.as_tensor(pts4, dtype=torch.float32)
    orc.point_inertia = lambda mass, P: _pi(mass, P4.to(P).unsqueeze(0))

    def oracle(dt):
        zd = z.to(dt).requires_grad_(True); cd = ctrl.to(dt).requires_grad_(True)
        md = None if mu is None else mu.to(dt).requires_grad_(True)
        sd = None if state is None else tuple(s.to(dt) for s in state)
        rs, rf = orc.rollout(spec, ex(zd), cd, state=sd, friction=ex(md))
        loss_of(list(rs) + list(rf), dt).backward()
        return [zd.grad, cd.grad] + ([md.grad] if md is not None else [])
    try:
        r64, r32 = oracle(torch.float64), oracle(torch.float32)
    finally:
        orc.point_inertia = _pi
    res.append(judge('cp', seed, info, got, r64, r32))
if res:
    summary('component-parallel kernels (<= 4 points) vs the float64 oracle', res)

# ---- mw -------------------------------------------------------------------------------------------------------------------------
res = []
for seed in range(SEED0, SEED0 + n_mw):
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
    info = dict(N=N, B=B, T=T, nt=nt, integ=integ, shared=shared, mu=use_mu, res=rs_, where=str(where), xs=xs_only)
    ex = lambda m: None if m is None else (m.expand(B, -1, -1) if m.shape[0] == 1 else m)  # noqa: E731

    def grads(fn, dev, dt):
        zl, cl = z.clone().to(dt).to(dev).requires_grad_(True), ctrl.clone().to(dt).to(dev).requires_grad_(True)
        ml = mu.clone().to(dt).to(dev).requires_grad_(True) if use_mu else None
        st = [t.clone().to(dt).to(dev) for t in state]
        for t in st[1:]:
            t.requires_grad_(True)
        outs = fn(ex(zl), cl, ex(ml), tuple(st))
        loss = (outs[0][:, ::3] * syn.probe_weights(outs[0][:, ::3].shape, 0.3, dtype=dt).to(dev)).sum() if xs_only else hp.probe_loss(outs, dt)
        loss.backward()
        return [(torch.zeros_like(g) if g.grad is None else g.grad).cpu() for g in [zl, cl] + ([ml] if use_mu else []) + st[1:]]      # (an input the loss does not reach: autograd leaves None)
    dp = make_dphysics(pts, masks, integ, rs_, d_max)
    spec = hp.spec_from(pts, masks, integ, rs_, d_max)

    def f_hip(zz, cc, mm, st):
        so, fo = dp(zz, cc, state=st, friction=mm)
        return list(so) + list(fo)

    def f_orc(zz, cc, mm, st):
        so, fo = orc.rollout(spec, zz, cc, state=st, friction=mm)
        return list(so) + list(fo)
    got = grads(f_hip, DEV, torch.float32)
    r64, r32 = grads(f_orc, 'cpu', torch.float64), grads(f_orc, 'cpu', torch.float32)
    res.append(judge('mw', seed, info, got, r64, r32))
if res:
    summary('recording forward + record-reading backward (5..300 points) vs the float64 oracle', res)
