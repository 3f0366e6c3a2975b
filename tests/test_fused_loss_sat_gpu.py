"""`physics_loss` (losses.py:102-127) INSIDE the saturated backward launches (round 6; rollout_bwd_kernel.h LOSS): beyond
8192 rollouts of the 4-point body -- and beyond the record-reading range of 5..64-point bodies -- the positions-only one-point-per-lane kernels
form dL/dXs at the stamped rows themselves.  Refereed by the float64 ORACLE: the ground truth of every rollout outside a small subset is set to
that rollout's OWN predicted positions (difference exactly zero: no gradient from it), so the oracle's autograd over the subset is the whole
gradient; and held against the unfused HIP route (dense dL/dXs rows through mf_physics_loss_*) on random ground truth for ALL rollouts."""
import ctypes as C

import pytest
import torch

from oracle import dphysics_oracle as orc
from tests import helpers as hp
from tests.test_rollout_gpu import make_dphysics

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _scatter_state(B, ctrl, seed=11):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.zeros(B, 3); x0[:, :2] = (torch.rand(B, 2, generator=g) - 0.5) * 12.6
    yaw = torch.rand(B, generator=g) * 6.2831853
    R0 = torch.zeros(B, 3, 3); R0[:, 0, 0] = yaw.cos(); R0[:, 0, 1] = -yaw.sin(); R0[:, 1, 0] = yaw.sin(); R0[:, 1, 1] = yaw.cos(); R0[:, 2, 2] = 1.0
    xd0 = torch.zeros(B, 3); xd0[:, 0] = ctrl[:, 0, 0] * yaw.cos(); xd0[:, 1] = ctrl[:, 0, 0] * yaw.sin()
    w0 = torch.zeros(B, 3); w0[:, 2] = ctrl[:, 0, 1]
    return x0, xd0, R0, w0


def _fusable(dp, B, T, H):
    from monoforce_amd import _lib
    d = _lib.MfRolloutDesc(B=B, T=T, N=dp.x_points.shape[1], H=H, W=H, integrator=_lib.MF_INTEG_ODEINT_EULER if dp.dphys_cfg.use_odeint else _lib.MF_INTEG_DYNAMICS,
                           math_mode=_lib.MF_MATH_FAST, map_shared=1, layout=_lib.MF_LAYOUT_TIME_MAJOR)
    return int(_lib.lib().mf_rollout_loss_fusable(C.byref(d)))


@pytest.mark.parametrize('integ', [1, 0])
@pytest.mark.parametrize('scattered', [False, True])
@pytest.mark.parametrize('B', [16384, 32768, 16384 + 192])
def test_fused_loss_in_the_saturated_backward_vs_float64_oracle(integ, scattered, B):
    from monoforce_amd import synthetic as syn, _timing
    if B == 16384 + 192 and (integ == 0 or scattered):
        pytest.skip('the ragged batch runs once')
    T, sub, every = 60, 24, 10
    pts, masks = syn.robot_points_4()
    z, mu = syn.bump_terrain(syn.bump_params(5), 6.4, 0.05), syn.wave_friction(6.4, 0.05)
    ctrl = syn.const_controls(B, T, seed=2)
    sel = torch.cat([torch.arange(0, B, B // (sub - 2))[:sub - 2], torch.tensor([B - 2, B - 1])])      # ... the last rollouts of a ragged wave too
    n = sel.numel()
    spec = hp.spec_from(pts, masks, integ, 0.05, 6.4)
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    assert _fusable(dp, B, T, 256) == 2
    state = _scatter_state(B, ctrl) if scattered else None
    dev_state = (lambda: tuple(t.clone().to(DEV) for t in state)) if scattered else (lambda: None)
    stamps = torch.arange(every - 1, T, every)
    full_ts = torch.linspace(0, 5.0, 500)[:T]
    lspec = dp.loss_spec(full_ts[stamps], gamma=0.9, n_steps=T)
    assert lspec.fusable and lspec.T2 == stamps.numel()
    with torch.no_grad():
        (X0, _, _, _), _ = dp(z.to(DEV).unsqueeze(0), ctrl.to(DEV), friction=mu.to(DEV).unsqueeze(0), state=dev_state())
    X_gt = X0[:, stamps.to(DEV)].contiguous().clone()                      # every rollout's own prediction: zero difference, zero gradient ...
    gen = torch.Generator().manual_seed(5)
    tgt = X_gt[sel.to(DEV)].cpu() + 0.2 * (torch.rand(n, stamps.numel(), 3, generator=gen) - 0.5)
    X_gt[sel.to(DEV)] = tgt.to(DEV)                                        # ... but for the subset the oracle differentiates
    zd, md, cd = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True), ctrl.to(DEV).requires_grad_(True)
    _timing.start()
    loss, (Xs, _, _, _) = dp.physics_loss_rollout(zd.unsqueeze(0), cd, X_gt, lspec, state=dev_state(), friction=md.unsqueeze(0))
    loss.backward()
    name = _timing.launches()['rollout_bwd_kernel']
    ks = _timing.stop()
    # rollout_bwd_kernel<float, 4, 1, INTEG, FAST, JOINTS, CARRY, XS_ONLY, ZMU, WIN, LOSS>: the fused instantiation ran, and no loss-gradient launch
    assert name.split(' grid')[0].endswith('true, true, true, true>') and 'rollout_bwd_kernel<float, 4, 1, %d,' % integ in name, name
    assert 'physics_loss_bwd' not in ks, ks.keys()
    count = B * stamps.numel() * 3

    def oracle(dtype):
        zc, mc, cc = z.to(dtype).requires_grad_(True), mu.to(dtype).requires_grad_(True), ctrl[sel].to(dtype).requires_grad_(True)
        st = tuple(t[sel].clone().to(dtype) for t in state) if scattered else None
        (rX, _, _, _), _ = orc.rollout(spec, zc.unsqueeze(0).expand(n, -1, -1), cc, state=st, friction=mc.unsqueeze(0).expand(n, -1, -1))
        w = (1. / (1. + 0.9 * full_ts[stamps].to(dtype))).view(1, -1, 1)
        lo = (((rX[:, stamps] * w - tgt.to(dtype) * w) ** 2).sum() / count)
        lo.backward()
        return lo.detach(), zc.grad, mc.grad, cc.grad
    ref, env = oracle(torch.float64), oracle(torch.float32)
    assert abs(float(loss) - float(ref[0])) <= max(2e-4, 3.0 * abs(float(env[0]) - float(ref[0])) / abs(float(ref[0]))) * abs(float(ref[0]))
    for nm, g_, r64, r32 in zip(('z', 'mu', 'controls'), (zd.grad, md.grad, cd.grad[sel.to(DEV)]), ref[1:], env[1:]):
        bar = max(2e-4, 3.0 * hp.rel_err(r32, r64))
        assert hp.rel_err(g_, r64) <= bar, (nm, hp.rel_err(g_, r64), 'bar', bar)
    rest = torch.ones(B, dtype=torch.bool); rest[sel] = False
    assert float(cd.grad[rest.to(DEV)].abs().max()) == 0.0                 # a rollout whose ground truth is its own prediction gets exactly nothing


@pytest.mark.parametrize('integ,B', [(1, 8192), (1, 4608), (1, 6144 + 4), (0, 8192), (0, 4100), (0, 6144)])
def test_fused_loss_in_the_one_wave_component_parallel_backward(integ, B):
    """The component-parallel backward in its EARLY-RECOMPUTE form (4097 .. 8192 rollouts, either integrator: no record, more than one wave per
    SIMD, cell gradients through the LDS window where the wave count allows) forms dL/dXs itself (mf_rollout_loss_fusable = 3) -- against the float64 ORACLE on a subset (every other rollout's ground truth = its own prediction) and
    against the unfused HIP route on random ground truth for all rollouts."""
    from monoforce_amd import synthetic as syn, _timing
    from monoforce_amd.losses import physics_loss_fused
    T, sub, every = 60, 16, 10
    pts, masks = syn.robot_points_4()
    z, mu = syn.bump_terrain(syn.bump_params(6), 6.4, 0.05), syn.wave_friction(6.4, 0.05)
    ctrl = syn.const_controls(B, T, seed=3)
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    assert _fusable(dp, B, T, 256) == 3
    sel = torch.cat([torch.arange(0, B, B // (sub - 2))[:sub - 2], torch.tensor([B - 2, B - 1])])
    n = sel.numel()
    spec = hp.spec_from(pts, masks, integ, 0.05, 6.4)
    stamps = torch.arange(every - 1, T, every)
    full_ts = torch.linspace(0, 5.0, 500)[:T]
    lspec = dp.loss_spec(full_ts[stamps], gamma=0.9, n_steps=T)
    with torch.no_grad():
        (X0, _, _, _), _ = dp(z.to(DEV).unsqueeze(0), ctrl.to(DEV), friction=mu.to(DEV).unsqueeze(0))
    gen = torch.Generator().manual_seed(7)
    X_own = X0[:, stamps.to(DEV)].contiguous()
    tgt = X_own[sel.to(DEV)].cpu() + 0.2 * (torch.rand(n, stamps.numel(), 3, generator=gen) - 0.5)
    X_sub = X_own.clone(); X_sub[sel.to(DEV)] = tgt.to(DEV)
    X_rand = (torch.rand(B, stamps.numel(), 3, generator=gen) - 0.5).to(DEV)
    count = B * stamps.numel() * 3

    def run(X_gt, route):
        zd, md, cd = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True), ctrl.to(DEV).requires_grad_(True)
        _timing.start()
        if route == 'fused':
            loss = dp.physics_loss_rollout(zd.unsqueeze(0), cd, X_gt, lspec, friction=md.unsqueeze(0))[0]
        else:
            keep, dp.return_forces = dp.return_forces, False
            states, _ = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0))
            dp.return_forces = keep
            loss = physics_loss_fused(states, [X_gt], None, lspec.gt_ts.unsqueeze(0).expand(B, -1), gamma=0.9, nearest=lspec.near.unsqueeze(0).expand(B, -1))
        loss.backward()
        name = _timing.launches()['rollout_bwd_kernel'].split(' grid')[0]
        ks = _timing.stop()
        assert 'rollout_bwd_cp_kernel<float, %d, true,' % integ in name, name
        assert ('physics_loss_bwd' in ks) == (route != 'fused'), (route, list(ks))
        return float(loss.detach()), zd.grad.clone(), md.grad.clone(), cd.grad.clone()
    a, b = run(X_rand, 'fused'), run(X_rand, 'unfused')
    assert abs(a[0] - b[0]) <= 2e-6 * abs(b[0])
    for k in (1, 2, 3):
        assert hp.rel_err(a[k], b[k]) <= 2e-5, (k, hp.rel_err(a[k], b[k]))
    got = run(X_sub, 'fused')

    def oracle(dtype):
        zc, mc, cc = z.to(dtype).requires_grad_(True), mu.to(dtype).requires_grad_(True), ctrl[sel].to(dtype).requires_grad_(True)
        (rX, _, _, _), _ = orc.rollout(spec, zc.unsqueeze(0).expand(n, -1, -1), cc, friction=mc.unsqueeze(0).expand(n, -1, -1))
        w = (1. / (1. + 0.9 * full_ts[stamps].to(dtype))).view(1, -1, 1)
        lo = (((rX[:, stamps] * w - tgt.to(dtype) * w) ** 2).sum() / count)
        lo.backward()
        return lo.detach(), zc.grad, mc.grad, cc.grad
    ref, env = oracle(torch.float64), oracle(torch.float32)
    assert abs(got[0] - float(ref[0])) <= max(2e-4, 3.0 * abs(float(env[0]) - float(ref[0])) / abs(float(ref[0]))) * abs(float(ref[0]))
    for nm, g_, r64, r32 in zip(('z', 'mu', 'controls'), (got[1], got[2], got[3][sel.to(DEV)]), ref[1:], env[1:]):
        bar = max(2e-4, 3.0 * hp.rel_err(r32, r64))
        assert hp.rel_err(g_, r64) <= bar, (nm, hp.rel_err(g_, r64), 'bar', bar)
    rest = torch.ones(B, dtype=torch.bool); rest[sel] = False
    assert float(got[3][rest.to(DEV)].abs().max()) == 0.0


@pytest.mark.parametrize('integ', [1, 0])
@pytest.mark.parametrize('B,N', [(16384, 4), (20480, 4), (16384 + 256, 8)])
def test_fused_loss_equals_the_unfused_hip_route(integ, B, N):
    """Random ground truth for ALL rollouts: the fused launch against forward + mf_physics_loss_value / _bwd + the backward reading dense rows
    (the same kernels with LOSS = false).  Same arithmetic per row; what differs is the order of the float atomics."""
    from monoforce_amd import synthetic as syn, _timing
    from monoforce_amd.losses import physics_loss_fused
    T, every = 50, 10
    pts, masks = syn.robot_points_4() if N == 4 else syn.robot_points_box(N, seed=5, n_tracks=2)
    z, mu = syn.bump_terrain(syn.bump_params(9), 6.4, 0.05), syn.wave_friction(6.4, 0.05)
    ctrl = syn.const_controls(B, T, seed=4).to(DEV)
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    assert _fusable(dp, B, T, 256) == 2
    stamps = torch.arange(every - 1, T, every)
    full_ts = torch.linspace(0, 5.0, 500)[:T]
    lspec = dp.loss_spec(full_ts[stamps], gamma=0.9, n_steps=T)
    gen = torch.Generator().manual_seed(3)
    X_gt = (torch.rand(B, stamps.numel(), 3, generator=gen) - 0.5).to(DEV)
    out = {}
    for route in ('fused', 'fused_value_in_backward', 'unfused'):
        zd, md = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True)
        _timing.start()
        if route.startswith('fused'):
            vib = route.endswith('backward')      # MF_LOSS_VALUE_IN_BACKWARD: NaN until the backward launch has formed it
            loss = dp.physics_loss_rollout(zd.unsqueeze(0), ctrl, X_gt, lspec, friction=md.unsqueeze(0), value_in_backward=vib)[0]
            assert bool(torch.isnan(loss.detach())) == vib
        else:
            keep, dp.return_forces = dp.return_forces, False
            states, _ = dp(zd.unsqueeze(0), ctrl, friction=md.unsqueeze(0))
            dp.return_forces = keep
            gt_ts = lspec.gt_ts.unsqueeze(0).expand(B, -1)
            loss = physics_loss_fused(states, [X_gt], None, gt_ts, gamma=0.9, nearest=lspec.near.unsqueeze(0).expand(B, -1))
        loss.backward()
        name = _timing.launches()['rollout_bwd_kernel'].split(' grid')[0]
        ks = _timing.stop()
        assert name.endswith(', true>' if route.startswith('fused') else ', false>'), (route, name)
        assert ('physics_loss_bwd' in ks) == (route == 'unfused') and ('physics_loss_fwd' in ks) == (route != 'fused_value_in_backward')
        out[route] = (float(loss.detach()), zd.grad.clone(), md.grad.clone())
    for route in ('fused', 'fused_value_in_backward'):
        assert abs(out[route][0] - out['unfused'][0]) <= 2e-6 * abs(out['unfused'][0]), (route, out[route][0], out['unfused'][0])
        for k in (1, 2):
            assert hp.rel_err(out[route][k], out['unfused'][k]) <= 2e-5, (route, k, hp.rel_err(out[route][k], out['unfused'][k]))


def test_fit_step_at_16384_rollouts_takes_the_fused_route_and_drops_the_dense_gradient():
    """`TerrainFitProblem` (bench.py's sweep, scripts/fit_terrain.py:53-62 at scale): forward + the fused backward (which forms the loss value
    too) + the reduction of the gradient copies -- no loss launch at all; launch by launch and replayed as a hipGraph give the same loss and gradients."""
    from monoforce_amd import synthetic as syn, _timing
    from monoforce_amd.train import TerrainFitProblem
    from bench import build_problem
    B, T = 16384, 100
    _, dp, _, _, z, mu, ctrl = build_problem(B, T, 4, DEV, 1, seed=0)
    prob = TerrainFitProblem(dp, syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(DEV), mu.to(DEV), ctrl.to(DEV), graph=True)
    zl, ml = z.to(DEV).clone().requires_grad_(True), mu.to(DEV).clone().requires_grad_(True)
    _timing.start()
    l0 = float(prob.step(zl, ml, eager=True))
    name = _timing.launches()['rollout_bwd_kernel'].split(' grid')[0]
    ks = _timing.stop()
    assert name.endswith('true, true, true, true>'), name      # XS_ONLY, ZMU, WIN, LOSS
    assert 'physics_loss_bwd' not in ks and 'physics_loss_fwd' not in ks and l0 == l0 and l0 > 0
    g0 = (zl.grad.clone(), ml.grad.clone())
    assert float(g0[0].abs().max()) > 0 and float(g0[1].abs().max()) > 0
    for _ in range(3):
        l1 = float(prob.step(zl, ml))                            # captured at the first call, replays afterwards
        assert abs(l1 - l0) <= 1e-6 * abs(l0)
        for a, b in zip((zl.grad, ml.grad), g0):
            assert hp.rel_err(a, b) <= 2e-5


@pytest.mark.parametrize('integ', [1, 0])
@pytest.mark.parametrize('N,B,ppl', [(100, 1100, 2), (223, 600, 4), (300, 520, 8)])
def test_positions_only_backward_with_several_points_per_lane_vs_oracle(integ, N, B, ppl):
    """Bodies of 65 .. 512 points beyond the record-reading multi-wave range (more than two waves per SIMD at one rollout over 2 / 4 / 8
    waves): one rollout per wave with 2 / 4 / 8 points per lane.  Round 6: a positions-only upstream (what `physics_loss` sends) runs the
    XS_ONLY instantiation there too (bench.py's points_sweep had these launches at 3.8 x their forward's time on the general kernel) --
    against the float64 ORACLE on the rollouts the loss touches."""
    from monoforce_amd import synthetic as syn, _timing
    T, sub = 24, 6
    pts, masks = syn.robot_points_box(N, seed=7, n_tracks=2)
    z, mu = syn.bump_terrain(syn.bump_params(4), 6.4, 0.1) * 0.5, syn.wave_friction(6.4, 0.1)
    ctrl = syn.const_controls(B, T, seed=6)
    sel = torch.cat([torch.arange(0, B, B // (sub - 1))[:sub - 1], torch.tensor([B - 1])])
    spec = hp.spec_from(pts, masks, integ, 0.1, 6.4)
    wts = syn.probe_weights((sel.numel(), T, 3), phase=0.4)
    dp = make_dphysics(pts, masks, integ, 0.1, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    zd, md, cd = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True), ctrl.to(DEV).requires_grad_(True)
    _timing.start()
    (Xs, _, _, _), _ = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0))
    (Xs[sel.to(DEV)] * wts.to(DEV)).sum().backward()
    name = _timing.launches()['rollout_bwd_kernel'].split(' grid')[0]
    _timing.stop()
    assert 'rollout_bwd_kernel<float, 64, %d, %d, true, false, true, true, false, false, false>' % (ppl, integ) in name, name      # CARRY, XS_ONLY

    def oracle_grads(dtype):
        zc, mc, cc = z.to(dtype).requires_grad_(True), mu.to(dtype).requires_grad_(True), ctrl[sel].to(dtype).requires_grad_(True)
        n = sel.numel()
        (rX, _, _, _), _ = orc.rollout(spec, zc.unsqueeze(0).expand(n, -1, -1), cc, friction=mc.unsqueeze(0).expand(n, -1, -1))
        (rX * wts.to(dtype)).sum().backward()
        return zc.grad, mc.grad, cc.grad
    ref, env = oracle_grads(torch.float64), oracle_grads(torch.float32)
    for nm, g_, r64, r32 in zip(('z', 'mu', 'controls'), (zd.grad, md.grad, cd.grad[sel.to(DEV)]), ref, env):
        bar = max(2e-3, 3.0 * hp.rel_err(r32, r64))      # (large bodies: the multi-wave tests' bar)
        assert hp.rel_err(g_, r64) <= bar, (nm, hp.rel_err(g_, r64), 'bar', bar)
    rest = torch.ones(B, dtype=torch.bool); rest[sel] = False
    assert float(cd.grad[rest.to(DEV)].abs().max()) == 0.0
