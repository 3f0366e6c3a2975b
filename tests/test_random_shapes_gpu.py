"""Seeded random problem shapes: HIP (float64, through the C ABI) vs the CPU oracle, forward and gradients.

The fixed-shape tests pin the reference's numbers; this sweep varies what they hold constant -- batch size, horizon, number
of contact points, map size and resolution, integrator, track count, friction map or none, shared or per-rollout maps, a
given start state or the default one, flat / bumpy / off-map starts -- to catch indexing and masking slips."""
import os
import numpy as np
import pytest
import torch

from tests import helpers as hp
from tests.test_rollout_gpu import make_dphysics

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _case(seed):
    from monoforce_amd import synthetic as syn
    rng = np.random.RandomState(1000 + seed)
    N = int(rng.choice([3, 4, 5, 9, 16, 17, 31, 40, 64, 65, 100, 129, 200]))
    n_tracks = int(rng.choice([2, 4]))
    B = int(rng.randint(1, 41))
    T = int(rng.randint(2, 61))
    H = int(rng.randint(10, 49))
    res = float(rng.choice([0.05, 0.1, 0.2]))
    d_max = H * res / 2
    integ = int(rng.randint(0, 2))
    shared = bool(rng.randint(0, 2))
    with_mu = bool(rng.randint(0, 3))
    with_state = bool(rng.randint(0, 2))
    pts, masks = syn.robot_points_box(N, seed=seed, n_tracks=n_tracks)
    nb = 1 if shared else B
    amp = float(rng.choice([0.0, 0.2, 0.6]))
    z = torch.stack([syn.bump_terrain(syn.bump_params(seed * 7 + b), d_max, res, torch.float64)[:H, :H] * amp for b in range(nb)])
    mu = torch.stack([syn.wave_friction(d_max, res, 0.4, 1.0, 1.0 + 0.3 * b, 0.7, torch.float64)[:H, :H] for b in range(nb)]) if with_mu else None
    ctrl = syn.varying_controls(B, T, seed=seed, dtype=torch.float64)
    state = None
    if with_state:
        from scipy.spatial.transform import Rotation
        f = lambda a: torch.as_tensor(a, dtype=torch.float64)  # noqa: E731
        span = d_max * float(rng.choice([0.3, 1.2]))       # 1.2: some rollouts start off the map (index clamping)
        state = (f(rng.uniform(-span, span, (B, 3)) * [1, 1, 0.05]), f(rng.uniform(-0.5, 0.5, (B, 3))),
                 f(Rotation.from_euler('xyz', rng.uniform(-0.2, 0.2, (B, 3)) * [1, 1, 15]).as_matrix()), f(rng.uniform(-0.3, 0.3, (B, 3))))
    return dict(N=N, n_tracks=n_tracks, B=B, T=T, H=H, res=res, d_max=d_max, integ=integ, shared=shared), pts, masks, z, mu, ctrl, state


@pytest.mark.parametrize('seed', range(24))
def test_random_shape_forward_and_gradients_vs_oracle_f64(seed):
    from oracle import dphysics_oracle as orc
    info, pts, masks, z, mu, ctrl, state = _case(seed)
    B, T = info['B'], info['T']
    spec = hp.spec_from(pts, masks, info['integ'], info['res'], info['d_max'])
    expand = lambda m: None if m is None else (m.expand(B, -1, -1) if info['shared'] else m)  # noqa: E731

    def run(fn, dev):
        zl = z.clone().to(dev).requires_grad_(True)
        cl = ctrl.clone().to(dev).requires_grad_(True)
        ml = None if mu is None else mu.clone().to(dev).requires_grad_(True)
        st = None
        if state is not None:
            st = [s.clone().to(dev) for s in state]
            for s in st[1:]:
                s.requires_grad_(True)
        outs = fn(expand(zl), cl, None if st is None else tuple(st), expand(ml))
        hp.probe_loss(outs, torch.float64).backward()
        grads = [zl.grad, cl.grad] + ([] if ml is None else [ml.grad]) + ([] if st is None else [s.grad for s in st[1:]])
        x0_after = None if st is None else st[0].detach().cpu()
        return [o.detach().cpu() for o in outs], [g.cpu() for g in grads], x0_after

    def f_oracle(zz, cc, st, mm):
        so, fo = orc.rollout(spec, zz, cc, state=st, friction=mm)
        return list(so) + list(fo)

    dp = make_dphysics(pts, masks, info['integ'], info['res'], info['d_max'])      # default time grid linspace(0, 5, 500)[:T]

    def f_hip(zz, cc, st, mm):
        s, f = dp(zz, cc, state=st, friction=mm)
        return list(s) + list(f)

    o_ref, g_ref, x_ref = run(f_oracle, 'cpu')
    o_hip, g_hip, x_hip = run(f_hip, DEV)
    for k, a, b in zip(hp.OUT_KEYS, o_hip, o_ref):
        assert a.shape == b.shape, (info, k, a.shape, b.shape)
        assert hp.rel_err(a, b) <= 1e-9, (info, k, hp.rel_err(a, b))
    for i, (a, b) in enumerate(zip(g_hip, g_ref)):
        assert hp.rel_err(a, b) <= 1e-7, (info, 'grad', i, hp.rel_err(a, b))
    if x_ref is not None:
        assert hp.rel_err(x_hip, x_ref) <= 1e-12       # the terrain snap written back into the caller's x0


@pytest.mark.parametrize('precise', [False, True])
@pytest.mark.parametrize('seed', range(12))
def test_random_shape_forward_f32_vs_oracle_f64(seed, precise):
    """float32 kernels (fast and exact arithmetic) on the same random shapes: <= 1e-4 relative on poses (north_star bar) at
    these horizons (T <= 60), 1e-3 on the force rows (they carry the contact model's amplification of pose rounding)."""
    from oracle import dphysics_oracle as orc
    info, pts, masks, z, mu, ctrl, state = _case(seed)
    B = info['B']
    spec = hp.spec_from(pts, masks, info['integ'], info['res'], info['d_max'])
    expand = lambda m: None if m is None else (m.expand(B, -1, -1) if info['shared'] else m)  # noqa: E731
    with torch.no_grad():
        so, fo = orc.rollout(spec, expand(z), ctrl, state=None if state is None else tuple(s.clone() for s in state), friction=expand(mu))
        dp = make_dphysics(pts, masks, info['integ'], info['res'], info['d_max'], precise=precise)
        f32 = lambda t: None if t is None else t.to(torch.float32).to(DEV)  # noqa: E731
        sh, fh = dp(expand(f32(z)), f32(ctrl), state=None if state is None else tuple(f32(s) for s in state), friction=expand(f32(mu)))
    for k, a, b in zip(hp.OUT_KEYS, list(sh) + list(fh), list(so) + list(fo)):
        tol = 1e-4 if k in ('Xs', 'Rs') else 1e-3
        assert hp.rel_err(a.cpu().double(), b) <= tol, (info, k, hp.rel_err(a.cpu().double(), b))


@pytest.mark.parametrize('B,N,n_tracks', [(3, 100, 2), (5, 223, 4), (2, 400, 2), (600, 100, 2), (300, 200, 4), (260, 300, 2)])
def test_large_body_lane_mappings_vs_oracle_f64(B, N, n_tracks):
    """Bodies of more than 64 points: a small batch spreads ONE rollout over 2 / 4 / 8 waves (LDS exchange between them), a
    large one keeps one wave per rollout with 2 / 4 / 8 points per lane -- forward and gradients of both against the oracle."""
    from monoforce_amd import synthetic as syn
    from oracle import dphysics_oracle as orc
    pts, masks = syn.robot_points_box(N, seed=N + B, n_tracks=n_tracks)
    T = 10 if B > 100 else 25
    z1 = syn.bump_terrain(syn.bump_params(B), 3.2, 0.1, torch.float64) * 0.4
    mu1 = syn.wave_friction(3.2, 0.1, 0.5, 1.0, 1.4, 0.8, torch.float64)
    ctrl = syn.varying_controls(B, T, seed=B, dtype=torch.float64)
    for integ in (0, 1):
        spec = hp.spec_from(pts, masks, integ, 0.1, 3.2)

        def run(fn, dev):
            zl, ml, cl = (t.clone().to(dev).requires_grad_(True) for t in (z1, mu1, ctrl))
            outs = fn(zl.unsqueeze(0).expand(B, -1, -1), cl, ml.unsqueeze(0).expand(B, -1, -1))
            hp.probe_loss(outs, torch.float64).backward()
            return [o.detach().cpu() for o in outs], [g.grad.cpu() for g in (zl, ml, cl)]

        def f_oracle(zz, cc, mm):
            so, fo = orc.rollout(spec, zz, cc, friction=mm)
            return list(so) + list(fo)

        dp = make_dphysics(pts, masks, integ, 0.1, 3.2)

        def f_hip(zz, cc, mm):
            s, f = dp(zz, cc, friction=mm)
            return list(s) + list(f)

        o_ref, g_ref = run(f_oracle, 'cpu')
        o_hip, g_hip = run(f_hip, DEV)
        for k, a, b in zip(hp.OUT_KEYS, o_hip, o_ref):
            assert hp.rel_err(a, b) <= 1e-9, (B, N, integ, k, hp.rel_err(a, b))
        for nm, a, b in zip(('z', 'mu', 'controls'), g_hip, g_ref):
            assert hp.rel_err(a, b) <= 1e-7, (B, N, integ, nm, hp.rel_err(a, b))


@pytest.mark.parametrize('tag', ['f32', 'f64'])
def test_articulated_large_body_small_batch_vs_oracle(tag):
    """robot 'marv' with moving flippers, 130 contact points, 2 rollouts: the articulated kernels on the multi-wave mapping."""
    from monoforce_amd import synthetic as syn
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    from oracle import dphysics_oracle as orc
    dt = hp.DT[tag]
    B, T, N = 2, 20, 130
    pts, masks = syn.robot_points_box(N, seed=3, n_tracks=4)
    cfg = DPhysConfig(robot='marv', grid_res=0.1, robot_points=pts, driving_parts=masks)
    cfg.robot_mass = 40.0
    cfg.damping = float(np.sqrt(4 * cfg.robot_mass * cfg.stiffness))
    cfg.d_max = 1.6
    z = torch.stack([syn.bump_terrain(np.array([[0.12, 0.5, 0.2, 0.6]]), 1.6, 0.1, torch.float64) + 0.01 * k for k in range(B)])
    ctrl = syn.varying_controls(B, T, seed=6, dtype=torch.float64)
    t = torch.linspace(0, 1, T, dtype=torch.float64).view(1, T, 1)
    ja = 0.5 * torch.sin(2 * np.pi * (t * torch.tensor([1.0, 0.7, 1.3, 0.5]) + torch.arange(B).view(B, 1, 1) * 0.2))
    for integ in (0, 1):
        cfg.use_odeint = (integ == 1)
        spec = hp.spec_from(pts, masks, integ, 0.1, 1.6)
        spec.joint_positions = [list(v) for v in cfg.joint_positions.values()]
        zo = z.clone().requires_grad_(True)
        so, fo = orc.rollout(spec, zo, ctrl, joint_angles=ja)
        hp.probe_loss(list(so) + list(fo), torch.float64).backward()
        dp = DPhysics(cfg, device=DEV)
        zh = z.to(dt).to(DEV).requires_grad_(True)
        sh, fh = dp(zh, ctrl.to(dt).to(DEV), joint_angles=ja.to(dt).to(DEV))
        hp.probe_loss(list(sh) + list(fh), dt).backward()
        tol, gtol = (1e-9, 1e-7) if tag == 'f64' else (1e-4, 5e-4)
        for k, a, b in zip(hp.OUT_KEYS, list(sh) + list(fh), list(so) + list(fo)):
            assert hp.rel_err(a.detach().cpu().double(), b.detach()) <= tol, (integ, k, hp.rel_err(a.detach().cpu().double(), b.detach()))
        assert hp.rel_err(zh.grad.cpu().double(), zo.grad) <= gtol, (integ, hp.rel_err(zh.grad.cpu().double(), zo.grad))


@pytest.mark.parametrize('B,N', [(3, 64), (3, 223), (2, 300), (2, 400)])
def test_whole_wave_groups_f32_gradients_vs_oracle(B, N):
    """float32 kernels whose rollout group is a whole wave (or several): the scalar-register form of the wave reduction
    (row_bcast + readlane) in the forward and in the 23-component batched reduction of the backward.  The comparison is
    with the oracle in float32 -- the reference's own precision; rounding the inputs to float32 alone moves these
    contact-rich rollouts by 2e-4 relative to a float64 run."""
    from monoforce_amd import synthetic as syn
    from oracle import dphysics_oracle as orc
    pts, masks = syn.robot_points_box(N, seed=N, n_tracks=2)
    T = 30
    z = (syn.bump_terrain(syn.bump_params(20), 3.2, 0.1, torch.float64) * 0.3).unsqueeze(0).float()
    ctrl = syn.varying_controls(B, T, seed=N, dtype=torch.float64).float()
    for integ in (0, 1):
        spec = hp.spec_from(pts, masks, integ, 0.1, 3.2)
        zo = z.clone().requires_grad_(True)
        so, fo = orc.rollout(spec, zo.expand(B, -1, -1), ctrl)
        hp.probe_loss(list(so) + list(fo), torch.float32).backward()
        dp = make_dphysics(pts, masks, integ, 0.1, 3.2)
        zh = z.to(DEV).requires_grad_(True)
        sh, fh = dp(zh.expand(B, -1, -1), ctrl.to(DEV))
        hp.probe_loss(list(sh) + list(fh), torch.float32).backward()
        for k, a, b in zip(hp.OUT_KEYS, list(sh) + list(fh), list(so) + list(fo)):
            # two float32 evaluation orders of these contact-rich rollouts sit up to 1-2e-4 apart themselves
            # (per-point forces are k * penetration with k = 5e4: a float32 ulp of height is 1e-2 of a 2-newton force)
            tol = 5e-4 if k in ('Xs', 'Rs') else (3e-2 if k in ('Fs', 'Ff') else 2e-3)
            assert hp.rel_err(a.detach().cpu(), b.detach()) <= tol, (N, integ, k, hp.rel_err(a.detach().cpu(), b.detach()))
        assert hp.rel_err(zh.grad.cpu(), zo.grad) <= 2e-3, (N, integ, hp.rel_err(zh.grad.cpu(), zo.grad))


def _cp_case(seed):
    """Random small-body problems for the component-parallel kernels (float32 fast math, N <= 4 contact points)."""
    from monoforce_amd import synthetic as syn
    rng = np.random.RandomState(5000 + seed)
    N = int(rng.randint(1, 5))
    pts4, _ = syn.robot_points_4()
    pts = pts4[:N].copy()
    if N < 3:
        pts[:, 2] += np.array([0.0, 0.05])[:N]
    masks = [pts[:, 1] > 0, pts[:, 1] <= 0]
    B = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 33, 64, 130]))
    T = int(rng.randint(1, 70))
    H = int(rng.randint(12, 49))
    res = float(rng.choice([0.05, 0.1, 0.2]))
    d_max = H * res / 2
    integ = int(rng.randint(0, 2))
    shared = bool(rng.randint(0, 2))
    with_mu = bool(rng.randint(0, 3))
    with_state = bool(rng.randint(0, 2))
    loss_kind = int(rng.randint(0, 3))          # all outputs / positions only / positions + forces
    nb = 1 if shared else B
    amp = float(rng.choice([0.0, 0.2, 0.5]))
    z = torch.stack([syn.bump_terrain(syn.bump_params(seed * 5 + b), d_max, res)[:H, :H] * amp for b in range(nb)])
    mu = torch.stack([syn.wave_friction(d_max, res, 0.4, 1.0, 1.0 + 0.3 * b, 0.7)[:H, :H] for b in range(nb)]) if with_mu else None
    ctrl = syn.varying_controls(B, T, seed=seed)
    state = None
    if with_state:
        from scipy.spatial.transform import Rotation
        f = lambda a: torch.as_tensor(a, dtype=torch.float32)  # noqa: E731
        span = d_max * float(rng.choice([0.3, 1.2]))
        state = (f(rng.uniform(-span, span, (B, 3)) * [1, 1, 0.05]), f(rng.uniform(-0.5, 0.5, (B, 3))),
                 f(Rotation.from_euler('xyz', rng.uniform(-0.2, 0.2, (B, 3)) * [1, 1, 15]).as_matrix()), f(rng.uniform(-0.3, 0.3, (B, 3))))
    return dict(N=N, B=B, T=T, H=H, res=res, integ=integ, shared=shared, mu=with_mu, state=with_state, loss=loss_kind), pts, masks, z, mu, ctrl, state, d_max


def _cp_vs_lanes(seed):
    from monoforce_amd import synthetic as syn
    info, pts, masks, z, mu, ctrl, state, d_max = _cp_case(seed)
    B = info['B']
    pts4, _ = syn.robot_points_4()
    base = make_dphysics(pts4, [pts4[:, 1] > 0, pts4[:, 1] <= 0], info['integ'], info['res'], d_max)
    res = {}
    for ppl in (16, 1):
        dp = make_dphysics(pts4, [pts4[:, 1] > 0, pts4[:, 1] <= 0], info['integ'], info['res'], d_max, points_per_lane=ppl)
        dp.dphys_cfg.robot_points = torch.as_tensor(pts)
        dp.dphys_cfg.driving_parts = [torch.as_tensor(m) for m in masks]
        dp.x_points = dp.dphys_cfg.robot_points.unsqueeze(0).to(dp.device)
        dp._cache = {('iinv', torch.float32): base._iinv(torch.float32)}      # the 4-point body's inertia for both mappings
        zl = z.clone().to(DEV).requires_grad_(True)
        cl = ctrl.clone().to(DEV).requires_grad_(True)
        ml = None if mu is None else mu.clone().to(DEV).requires_grad_(True)
        st = None
        if state is not None:
            st = [s.clone().to(DEV) for s in state]
            for s in st[1:]:
                s.requires_grad_(True)
        ex = lambda m: None if m is None else (m.expand(B, -1, -1) if info['shared'] and B > 1 else m)  # noqa: E731
        so, fo = dp(ex(zl), cl, state=None if st is None else tuple(st), friction=ex(ml))
        outs = list(so) + list(fo)
        if info['loss'] == 0:
            loss = hp.probe_loss(outs, torch.float32)
        elif info['loss'] == 1:
            loss = (outs[0] * syn.probe_weights(outs[0].shape, phase=0.4).to(DEV)).sum()
        else:
            loss = (outs[0] * syn.probe_weights(outs[0].shape, phase=0.4).to(DEV)).sum() + 1e-3 * (outs[4] * syn.probe_weights(outs[4].shape, phase=1.4).to(DEV)).sum()
        loss.backward()
        grads = [zl.grad, cl.grad] + ([] if ml is None else [ml.grad]) + ([] if st is None else [s.grad for s in st[1:]])
        res[ppl] = ([o.detach().cpu() for o in outs], [g.cpu() for g in grads])
    worst = 0.0
    for k, a_, b_ in zip(hp.OUT_KEYS, res[16][0], res[1][0]):
        assert a_.shape == b_.shape and torch.isfinite(a_).all(), (info, k)
        worst = max(worst, hp.rel_err(a_, b_))
        assert hp.rel_err(a_, b_) <= 2e-4, (info, k, hp.rel_err(a_, b_))
    for i, (a_, b_) in enumerate(zip(res[16][1], res[1][1])):
        scale = float(b_.abs().max())
        if scale == 0.0:
            assert float(a_.abs().max()) == 0.0, (info, 'grad', i)
            continue
        worst = max(worst, hp.rel_err(a_, b_))
        assert hp.rel_err(a_, b_) <= 5e-4, (info, 'grad', i, hp.rel_err(a_, b_))
    return worst


@pytest.mark.parametrize('seed', range(24))
def test_random_shape_component_parallel_vs_one_point_per_lane_f32(seed):
    """The 16-lanes-per-rollout kernels against the one-point-per-lane kernels (themselves held to the oracle above) on random
    small problems: 1..4 contact points, 1..130 rollouts (partial waves), 1..69 steps (every remainder of the unrolled loops),
    both integrators, shared or per-rollout maps, friction map or none, given (also off-map) or default start state, and the
    three kinds of loss the backward is specialised for -- outputs <= 2e-4, every gradient <= 5e-4 (float32 fast math, two
    summation orders)."""
    _cp_vs_lanes(seed)


def test_backward_routes_take_the_forwards_decisions():
    """the backward from the forward's record and the backward that recomputes from the saved state rows both
    evaluate the forward's own formulas (rollout_cp_common.h cp_*: one definition, explicit fused multiply-adds), so they meet every
    clamp and the |F_n| kink on the forward's side: gradients agree to summation order on random problems -- round 2's seed 1233 (a
    normal force passing through zero, 0.1-7 % apart then) among them."""
    sys_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import importlib.util
    spec = importlib.util.spec_from_file_location('soak_self_consistency', os.path.join(sys_path, 'tools', 'soak_self_consistency.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    worst, bad = mod.compare([1233, 1234, 1235] + list(range(40, 52)), tol=2e-5)
    assert not bad, bad
