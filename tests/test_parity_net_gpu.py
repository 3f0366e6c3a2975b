"""Parity cases the first round left thin: `interpolate_grid` fed directly, the BASELINE configurations at
their own shapes, float32 `dynamics()` over the full horizon, component-wise error metrics, and the large-batch shared-map
backward (private gradient copies, accumulator carry-over) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import dphysics_oracle as orc
from tests import helpers as hp
from tests.test_rollout_gpu import make_dphysics, run_hip

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def comp_rel_err(a, b, comp_axes=1):
    """Component-wise companion of helpers.rel_err: every component index of the trailing `comp_axes` axes (x / y / z of a
    vec3, each of the 9 entries of R) is judged against ITS OWN largest reference magnitude over the whole tensor --
    the small ones (z of Xs, the off-axis entries of Rs) no longer hide behind the largest entry.  Returns the worst ratio."""
    a, b = (np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, np.float64) for v in (a, b))
    nc = int(np.prod(a.shape[a.ndim - comp_axes:]))
    a2, b2 = a.reshape(-1, nc), b.reshape(-1, nc)
    scale = np.maximum(np.abs(b2).max(0), 1e-30)
    return float((np.abs(a2 - b2).max(0) / scale).max())


def probe_body(dp, dtype):
    """Turn `dp` into a ONE-point body at the body origin (test-only): with R = I the terrain snap of `DPhysics.dphysics`
    (dphysics.py:567-571) then IS interpolate_grid at the start position.  A single point mass has a singular inertia, so
    the body inertia handed to the kernel is set to the identity (no torque arises: r = 0)."""
    cfg = dp.dphys_cfg
    cfg.robot_points = torch.zeros(1, 3)
    cfg.driving_parts = [torch.ones(1, dtype=torch.bool), torch.zeros(1, dtype=torch.bool)]
    dp.x_points = cfg.robot_points.unsqueeze(0).to(dp.device)
    dp._cache = {('iinv', dtype): [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0]}
    return dp


@pytest.mark.parametrize('tag', ['f32', 'f64'])
@pytest.mark.parametrize('ppl', [0, 1, 4, 16])
@pytest.mark.parametrize('precise', [False, True])
def test_interpolate_grid_golden_through_the_snap(tag, ppl, precise):
    """SURVEY 8 row a5, directly: the 16 edge / out-of-range queries of tests/golden/interp.npz (values of the reference's
    own interpolate_grid) through the HIP kernels: heights via the snap of a one-point body, normals via the direction of
    the spring force of dynamics()' first step (dh = 0 after the snap, so F_spring = -damping (xd . n) n)."""
    g = hp.load('interp')
    dt = hp.DT[tag]
    res, d_max = float(g['grid_res']), float(g['d_max'])
    grid = torch.as_tensor(g[f'{tag}/grid'])                     # [2, 8, 8]
    qx, qy = g[f'{tag}/qx'][0], g[f'{tag}/qy'][0]                # the same 16 queries for both grids
    nq = qx.shape[0]
    pts, masks = __import__('monoforce_amd.synthetic', fromlist=['x']).robot_points_4()
    dp = probe_body(make_dphysics(pts, masks, 0, res, d_max, points_per_lane=ppl, precise=precise), dt)
    dp.dphys_cfg.robot_mass = 1e6                                # force clamps at +-m g: far away
    for gi in range(2):
        x0 = torch.zeros(nq, 3, dtype=dt)
        x0[:, 0], x0[:, 1] = torch.as_tensor(qx), torch.as_tensor(qy)
        xd = torch.zeros(nq, 3, dtype=dt); xd[:, 2] = -1.0
        st = (x0, xd, torch.eye(3, dtype=dt).repeat(nq, 1, 1), torch.zeros(nq, 3, dtype=dt))
        ctrl = torch.zeros(nq, 1, 2, dtype=dt)
        outs, st_dev = run_hip(dp, grid[gi:gi + 1].to(dt), ctrl, st, None)       # ONE shared 8x8 map
        z_hip = st_dev[0][:, 2].cpu().numpy().astype(np.float64)
        F = outs[4][:, 0, 0].numpy().astype(np.float64)
        n_hip = F / np.linalg.norm(F, axis=-1, keepdims=True)
        z_ref, n_ref = g[f'{tag}/z'][gi].astype(np.float64), g[f'{tag}/n'][gi].astype(np.float64)
        # (queries sitting ON a cell boundary included: since round 3 the fast-math kernels form the cell coordinate as the correctly
        # rounded quotient -- Mth::cell_coord -- so they truncate into the reference's cell like the exact ones)
        tol = 1e-12 if tag == 'f64' else 2e-6
        ok = (np.abs(z_hip - z_ref) <= tol * np.abs(z_ref).max()) & (np.abs(n_hip - n_ref).max(-1) <= (1e-10 if tag == 'f64' else 2e-6))
        assert ok.all(), (gi, np.nonzero(~ok)[0], z_hip - z_ref, np.abs(n_hip - n_ref).max(-1))


def _c1_problem(dtype):
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    z = syn.bump_terrain(syn.bump_params(0, smooth=True), 6.4, 0.1, torch.float64).to(dtype).unsqueeze(0)      # [1,128,128]
    mu = syn.wave_friction(6.4, 0.1, dtype=torch.float64).to(dtype).unsqueeze(0)
    ctrl = syn.const_controls(1, 200, seed=3, dtype=torch.float64).to(dtype)
    return pts, masks, z, mu, ctrl


@pytest.mark.parametrize('tag', ['f32', 'f64'])
@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('ppl', [0, 1])
def test_config1_single_rollout_forward_backward_vs_oracle(tag, integ, ppl):
    """BASELINE configs[0] at its own shape: ONE rollout, 4 contact points, 200 Euler steps, 128x128 map at 0.1 m --
    all six outputs and the gradients to terrain, friction and controls vs the oracle."""
    dt = hp.DT[tag]
    pts, masks, z, mu, ctrl = _c1_problem(dt)
    assert z.shape == (1, 128, 128) and ctrl.shape == (1, 200, 2)
    dp = make_dphysics(pts, masks, integ, 0.1, 6.4, points_per_lane=ppl)
    zd, md, cd = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True), ctrl.to(DEV).requires_grad_(True)
    states, forces = dp(zd, cd, friction=md)
    outs = list(states) + list(forces)
    hp.probe_loss(outs, dt).backward()
    spec = hp.spec_from(pts, masks, integ, 0.1, 6.4)
    zc, mc, cc = z.clone().requires_grad_(True), mu.clone().requires_grad_(True), ctrl.clone().requires_grad_(True)
    rs, rf = orc.rollout(spec, zc, cc, friction=mc)
    refs = list(rs) + list(rf)
    hp.probe_loss(refs, dt).backward()
    tol = 1e-9 if tag == 'f64' else 1e-4
    for k, o, r in zip(hp.OUT_KEYS, outs, refs):
        assert hp.rel_err(o, r) <= tol, (k, hp.rel_err(o, r))
        # ... and component by component (z of Xs, every entry of R on its own scale); the y components of this nearly straight
        # drive are ~1e-2 of the x ones, so float32 gets a 10x wider bar here
        ce = comp_rel_err(o, r, 2 if k == 'Rs' else 1)
        assert ce <= (1e-8 if tag == 'f64' else 1e-3), (k, ce)
    gtol = 1e-7 if tag == 'f64' else 2e-3          # 200-step BPTT in float32: the reference's own fp32-vs-fp64 is ~1e-4 here
    assert hp.rel_err(zd.grad, zc.grad) <= gtol, hp.rel_err(zd.grad, zc.grad)
    assert hp.rel_err(md.grad, mc.grad) <= gtol, hp.rel_err(md.grad, mc.grad)
    assert hp.rel_err(cd.grad, cc.grad) <= gtol, hp.rel_err(cd.grad, cc.grad)


@pytest.mark.parametrize('ppl', [0, 1, 4])
@pytest.mark.parametrize('precise', [False, True])
def test_dynamics_integrator_f32_full_horizon_on_smooth_terrain(ppl, precise):
    """float32 `dynamics()` (use_odeint=False) over all 500 steps on the smooth and the flat terrain of the full-horizon
    fixture, vs the reference's own float32 outputs.  The reference's float32 run is itself only reproducible to its
    fp32-vs-fp64 envelope (~1e-3 on the positions here: the semi-implicit scheme with re-normalised rotations amplifies
    rounding more than the default integrator), so the bar is max(1e-4, 4 x that envelope) per output."""
    g = hp.load('rollout_full')
    pts, masks, z, mu, ctrl = hp.full_inputs(torch.float32)
    dp = make_dphysics(pts, masks, 0, hp.FULL['grid_res'], hp.FULL['d_max'], points_per_lane=ppl, precise=precise)
    outs, _ = run_hip(dp, z, ctrl, None, mu)
    for k, o in zip(hp.OUT_KEYS[:4], outs[:4]):
        r32, r64 = g[f'f32/i0/{k}'][2:], g[f'f64/i0/{k}'][2:]
        env = hp.rel_err(r32, r64)
        err = hp.rel_err(o[2:], r32)
        assert err <= max(1e-4, 4 * env), (k, err, env)
        assert env <= 2e-2, (k, env)      # (measured: 1.2e-3 on Xs, 7.6e-3 on Xds -- dynamics() is not a 1e-4 float32 path even in the reference)


def test_full_horizon_f32_calm_floor():
    """The reproducible-prefix test of test_rollout_gpu.py demands `n_calm > 960`; pin what the fixture actually offers so a
    regression cannot hide behind the low floor: with the default integrator the reference is reproducible (envelope
    <= 1e-6) on at least 3000 state-rows, and every one of them is checked at 1e-4 here for the default lane mapping."""
    g = hp.load('rollout_full')
    pts, masks, z, mu, ctrl = hp.full_inputs(torch.float32)
    for integ, floor in ((1, 3000), (0, 1200)):
        dp = make_dphysics(pts, masks, integ, hp.FULL['grid_res'], hp.FULL['d_max'])
        outs, _ = run_hip(dp, z, ctrl, None, mu)
        n_calm = 0
        for k, o in zip(hp.OUT_KEYS[:4], outs[:4]):
            r32, r64 = g[f'f32/i{integ}/{k}'].astype(np.float64), g[f'f64/i{integ}/{k}']
            o = o.numpy().astype(np.float64)
            B, T = r32.shape[:2]
            scale = np.abs(r64).reshape(B, -1).max(1).clip(1e-30)[:, None]
            env = np.maximum.accumulate(np.abs(r32 - r64).reshape(B, T, -1).max(2) / scale, axis=1)
            err = np.abs(o - r32).reshape(B, T, -1).max(2) / scale
            calm = env <= 1e-6
            n_calm += int(calm.sum())
            assert (err[calm] <= 1e-4).all(), (integ, k, float(err[calm].max()))
        assert n_calm >= floor, (integ, n_calm)


@pytest.mark.parametrize('B,ppl,integ', [(1024, 1, 1), (4096, 1, 1), (1024, 0, 1), (2048, 0, 1), (8192, 0, 1), (1024, 0, 0), (4096, 0, 0), (8192, 0, 0)])
def test_large_batch_shared_map_backward_vs_oracle(B, ppl, integ):
    """The shared-map backward at BASELINE batch sizes -- private gradient copies (rollout b -> copy b % copies) and, from
    192 waves up, the accumulator carry-over kernels -- against the ORACLE: the loss touches 32 rollouts spread over the
    batch (every 32nd or 128th; the others run with zero upstream gradients), so the oracle only has to differentiate those.
    ppl = 0 is the library's own choice: the component-parallel kernels in their three forms (producer / consumer waves up
    to B = 1024, late recompute up to 4096, early up to 8192), both integrators."""
    from monoforce_amd import synthetic as syn
    T, sub = 100, 32
    pts, masks = syn.robot_points_4()
    z = syn.bump_terrain(syn.bump_params(5), 6.4, 0.05)
    mu = syn.wave_friction(6.4, 0.05)
    ctrl = syn.const_controls(B, T, seed=2)
    sel = torch.arange(0, B, B // sub)[:sub]
    spec = hp.spec_from(pts, masks, integ, 0.05, 6.4)
    wts = syn.probe_weights((sub, T, 3), phase=0.3)
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4, points_per_lane=ppl)
    dp.dphys_cfg.traj_sim_time = 5.0
    zd, md = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True)
    cd = ctrl.to(DEV).requires_grad_(True)
    (Xs, Xds, Rs, Om), _ = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0))
    ((Xs[sel.to(DEV)] * wts.to(DEV)).sum() + (Om[sel.to(DEV)] * wts.to(DEV)).sum() * 0.1).backward()
    # The ORACLE referees both integrators: its float64 autograd on the 32 rollouts the loss touches.  The bar is 2e-4 or, where the
    # oracle's OWN float32 run lands further than that from its float64 run, three times that distance: BPTT through 100 steps of stiff
    # contact amplifies float32 rounding, and rollouts on a contact switch are chaotic (SURVEY fact 6 -- dynamics() on this terrain: 9 %,
    # every float32 evaluation order alike).  What float32 cannot show there the float64 build of these kernels does: 1e-7 against the same
    # oracle at the same sizes (tests/test_cp_f64_validation_gpu.py::test_large_batch_shared_map_backward_f64_vs_oracle).
    def oracle_grads(dtype):
        zc, mc = z.to(dtype).requires_grad_(True), mu.to(dtype).requires_grad_(True)
        cc = ctrl[sel].to(dtype).requires_grad_(True)
        (rX, _, _, rO), _ = orc.rollout(spec, zc.unsqueeze(0).expand(sub, -1, -1), cc, friction=mc.unsqueeze(0).expand(sub, -1, -1))
        ((rX * wts.to(dtype)).sum() + (rO * wts.to(dtype)).sum() * 0.1).backward()
        return zc.grad, mc.grad, cc.grad
    ref, env = oracle_grads(torch.float64), oracle_grads(torch.float32)
    for nm, got, r64, r32 in zip(('z', 'mu', 'controls'), (zd.grad, md.grad, cd.grad[sel.to(DEV)]), ref, env):
        bar = max(2e-4, 3.0 * hp.rel_err(r32, r64))
        assert hp.rel_err(got, r64) <= bar, (nm, hp.rel_err(got, r64), 'bar', bar)
    rest = torch.ones(B, dtype=torch.bool); rest[sel] = False
    assert float(cd.grad[rest.to(DEV)].abs().max()) == 0.0          # rollouts the loss does not touch get exactly nothing


@pytest.mark.parametrize('integ', [1, 0])
@pytest.mark.parametrize('friction', [True, False])
@pytest.mark.parametrize('scattered', [False, True])
@pytest.mark.parametrize('B', [16384, 8192, 32768])
def test_saturated_positions_only_backward_vs_oracle(integ, friction, scattered, B):
    """B = 16 384 rollouts of the 4-point body (one wave of sixteen rollouts on every SIMD), loss on the positions only: the XS_ONLY
    instantiations of the general backward (round 5), reading the shared pair interleaved when there is a friction map (ZMU), cell
    gradients through the workgroup's 128 x 128-cell LDS window (WIN; MF_BWD_WIN=0: register accumulators + atomics) -- against the
    float64 ORACLE on the 32 rollouts the loss touches, at the bar derived from the oracle alone.  `scattered`: given start poses all
    over the map, so that most rollouts of a workgroup lie OUTSIDE its window (centred on its first rollout) and take the atomics."""
    import os
    from monoforce_amd import synthetic as syn, _timing
    T, sub = 60, 32      # (B = 8192: two waves per SIMD of the component-parallel early-recompute kernel, the same window)
    win = os.environ.get('MF_BWD_WIN', '1') != '0'
    if not win and not (friction and B in (8192, 16384)):
        pytest.skip('the register-accumulator child runs the eight cases with a friction map at 8192 / 16 384 rollouts (VERDICT r5 item 9)')
    pts, masks = syn.robot_points_4()
    z = syn.bump_terrain(syn.bump_params(5), 6.4, 0.05)
    mu = syn.wave_friction(6.4, 0.05) if friction else None
    ctrl = syn.const_controls(B, T, seed=2)
    sel = torch.arange(0, B, B // sub)[:sub]
    spec = hp.spec_from(pts, masks, integ, 0.05, 6.4)
    wts = syn.probe_weights((sub, T, 3), phase=0.3)
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    state = None
    if scattered:
        g = torch.Generator().manual_seed(11)
        x0 = torch.zeros(B, 3); x0[:, :2] = (torch.rand(B, 2, generator=g) - 0.5) * 12.6      # up to the map's edge (+-6.3 of +-6.4 m) and past the window
        yaw = torch.rand(B, generator=g) * 6.2831853
        R0 = torch.zeros(B, 3, 3); R0[:, 0, 0] = yaw.cos(); R0[:, 0, 1] = -yaw.sin(); R0[:, 1, 0] = yaw.sin(); R0[:, 1, 1] = yaw.cos(); R0[:, 2, 2] = 1.0
        xd0 = torch.zeros(B, 3); xd0[:, 0] = ctrl[:, 0, 0] * yaw.cos(); xd0[:, 1] = ctrl[:, 0, 0] * yaw.sin()
        w0 = torch.zeros(B, 3); w0[:, 2] = ctrl[:, 0, 1]
        state = (x0, xd0, R0, w0)
    zd = z.to(DEV).requires_grad_(True)
    md = mu.to(DEV).requires_grad_(True) if friction else None
    cd = ctrl.to(DEV).requires_grad_(True)
    _timing.start()
    (Xs, Xds, Rs, Om), _ = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0) if friction else None,
                              state=tuple(t.clone().to(DEV) for t in state) if scattered else None)
    (Xs[sel.to(DEV)] * wts.to(DEV)).sum().backward()
    ran = _timing.launches()
    _timing.stop()
    name = ran['rollout_bwd_kernel']
    if B >= 16384:      # rollout_bwd_kernel<float, 4, 1, INTEG, FAST, JOINTS, CARRY, XS_ONLY, ZMU, WIN, LOSS>
        # (with the window the accumulator carry-over runs from two waves per SIMD up only: 32 768 rollouts, eight-wave workgroups)
        assert 'rollout_bwd_kernel<float, 4, 1, %d, true, false, %s, true, %s, %s, false>' % (integ, 'false' if (win and B < 32768) else 'true', 'true' if friction else 'false', 'true' if win else 'false') in name, name
    else:               # rollout_bwd_cp_kernel<float, INTEG, XS_ONLY, GCTRL, MODE = early, SLOTS, BATCH, ZMU, WIN, ONE1>
        assert 'rollout_bwd_cp_kernel<float, %d, true, true, 0, 6, 3, false%s, false>' % (integ, ', true' if win else ', false') in name, name

    def oracle_grads(dtype):
        zc = z.to(dtype).requires_grad_(True)
        mc = mu.to(dtype).requires_grad_(True) if friction else None
        cc = ctrl[sel].to(dtype).requires_grad_(True)
        st = tuple(t[sel].clone().to(dtype) for t in state) if scattered else None
        (rX, _, _, _), _ = orc.rollout(spec, zc.unsqueeze(0).expand(sub, -1, -1), cc, state=st,
                                       friction=mc.unsqueeze(0).expand(sub, -1, -1) if friction else None)
        (rX * wts.to(dtype)).sum().backward()
        return (zc.grad, mc.grad, cc.grad) if friction else (zc.grad, cc.grad)
    ref, env = oracle_grads(torch.float64), oracle_grads(torch.float32)
    got = (zd.grad, md.grad, cd.grad[sel.to(DEV)]) if friction else (zd.grad, cd.grad[sel.to(DEV)])
    for nm, g_, r64, r32 in zip(('z', 'mu', 'controls') if friction else ('z', 'controls'), got, ref, env):
        bar = max(2e-4, 3.0 * hp.rel_err(r32, r64))
        assert hp.rel_err(g_, r64) <= bar, (nm, hp.rel_err(g_, r64), 'bar', bar)
    rest = torch.ones(B, dtype=torch.bool); rest[sel] = False
    assert float(cd.grad[rest.to(DEV)].abs().max()) == 0.0


@pytest.mark.parametrize('res,B', [(0.2, 16384 + 37), (0.1, 16384 + 4)])
def test_saturated_backward_small_maps_and_ragged_batches_vs_oracle(res, B):
    """The LDS-window kernel where its window is LARGER than the map (64 x 64 cells) or exactly the map (128 x 128), with a batch that
    leaves the last wave and the last workgroup partly empty -- the loss touches the last rollouts too -- against the float64 oracle."""
    from monoforce_amd import synthetic as syn, _timing
    T, sub = 40, 32
    pts, masks = syn.robot_points_4()
    z = syn.bump_terrain(syn.bump_params(7), 6.4, res)
    mu = syn.wave_friction(6.4, res)
    assert z.shape[-1] == int(round(12.8 / res))
    ctrl = syn.const_controls(B, T, seed=3)
    sel = torch.cat([torch.arange(0, B, B // (sub - 4))[:sub - 4], torch.arange(B - 4, B)])
    spec = hp.spec_from(pts, masks, 1, res, 6.4)
    wts = syn.probe_weights((sel.numel(), T, 3), phase=0.7)
    dp = make_dphysics(pts, masks, 1, res, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    zd, md, cd = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True), ctrl.to(DEV).requires_grad_(True)
    _timing.start()
    (Xs, _, _, _), _ = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0))
    (Xs[sel.to(DEV)] * wts.to(DEV)).sum().backward()
    name = _timing.launches()['rollout_bwd_kernel']
    _timing.stop()
    import os
    if os.environ.get('MF_BWD_WIN', '1') != '0':
        assert name.split(' grid')[0].endswith('true, true, false>'), name      # ZMU, WIN, no fused loss

    def oracle_grads(dtype):
        zc, mc, cc = z.to(dtype).requires_grad_(True), mu.to(dtype).requires_grad_(True), ctrl[sel].to(dtype).requires_grad_(True)
        n = sel.numel()
        (rX, _, _, _), _ = orc.rollout(spec, zc.unsqueeze(0).expand(n, -1, -1), cc, friction=mc.unsqueeze(0).expand(n, -1, -1))
        (rX * wts.to(dtype)).sum().backward()
        return zc.grad, mc.grad, cc.grad
    ref, env = oracle_grads(torch.float64), oracle_grads(torch.float32)
    for nm, g_, r64, r32 in zip(('z', 'mu', 'controls'), (zd.grad, md.grad, cd.grad[sel.to(DEV)]), ref, env):
        bar = max(2e-4, 3.0 * hp.rel_err(r32, r64))
        assert hp.rel_err(g_, r64) <= bar, (nm, hp.rel_err(g_, r64), 'bar', bar)


def test_saturated_positions_only_backward_of_an_eight_point_body_vs_oracle():
    """Beyond the multi-wave record kernels' range (two waves per SIMD) an 8-point body runs the positions-only general backward at eight lanes
    per rollout (XS_ONLY, un-summed adjoint state, device-scope atomics: no window beyond four lanes) -- against the float64 oracle."""
    from monoforce_amd import synthetic as syn, _timing
    B, T, sub, N = 16384 + 256, 40, 24, 8
    pts, masks = syn.robot_points_box(N, seed=5, n_tracks=2)
    z = syn.bump_terrain(syn.bump_params(9), 6.4, 0.05)
    mu = syn.wave_friction(6.4, 0.05)
    ctrl = syn.const_controls(B, T, seed=4)
    sel = torch.arange(0, B, B // sub)[:sub]
    spec = hp.spec_from(pts, masks, 1, 0.05, 6.4)
    wts = syn.probe_weights((sub, T, 3), phase=0.2)
    dp = make_dphysics(pts, masks, 1, 0.05, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    zd, md, cd = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True), ctrl.to(DEV).requires_grad_(True)
    _timing.start()
    (Xs, _, _, _), _ = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0))
    (Xs[sel.to(DEV)] * wts.to(DEV)).sum().backward()
    name = _timing.launches()['rollout_bwd_kernel']
    _timing.stop()
    assert 'rollout_bwd_kernel<float, 8, 1, 1, true, false, true, true, true, false, false>' in name, name      # CARRY, XS_ONLY, ZMU, no WIN, no fused loss

    def oracle_grads(dtype):
        zc, mc, cc = z.to(dtype).requires_grad_(True), mu.to(dtype).requires_grad_(True), ctrl[sel].to(dtype).requires_grad_(True)
        (rX, _, _, _), _ = orc.rollout(spec, zc.unsqueeze(0).expand(sub, -1, -1), cc, friction=mc.unsqueeze(0).expand(sub, -1, -1))
        (rX * wts.to(dtype)).sum().backward()
        return zc.grad, mc.grad, cc.grad
    ref, env = oracle_grads(torch.float64), oracle_grads(torch.float32)
    for nm, g_, r64, r32 in zip(('z', 'mu', 'controls'), (zd.grad, md.grad, cd.grad[sel.to(DEV)]), ref, env):
        bar = max(2e-4, 3.0 * hp.rel_err(r32, r64))
        assert hp.rel_err(g_, r64) <= bar, (nm, hp.rel_err(g_, r64), 'bar', bar)


def test_saturated_backward_without_the_lds_window_vs_oracle():
    """The same cases on the register-accumulator kernels (MF_BWD_WIN=0 is read once per process: a child runs them)."""
    import os, subprocess, sys
    if os.environ.get('MF_BWD_WIN') == '0':
        pytest.skip('this IS the child')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(repo, 'tests', 'test_parity_net_gpu.py'), '-x', '-q', '-m', 'gpu', '-k',
                        'test_saturated_positions_only_backward_vs_oracle'], env=dict(os.environ, MF_BWD_WIN='0'), capture_output=True, text=True,
                       timeout=1200, cwd=repo)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert '8 passed' in r.stdout and '16 skipped' in r.stdout, r.stdout[-500:]


def test_config4_full_size_step_vs_oracles():
    """BASELINE configs[3] at full size, one step: 4 cameras x 3x256x512 -> 256x256 BEV, 1024 rollouts x 500 steps on the
    predicted terrain.  BEV vs the splat oracle on the lifted features; rollout states vs the rollout oracle on the
    PREDICTED maps (a 128-rollout subset, float32 calm prefix + full-horizon boundedness); finite, non-zero gradients."""
    from oracle import splat_oracle as so
    from monoforce_amd import synthetic as syn
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    from monoforce_amd.train import EncoderTrainStep, synthetic_encoder_batch
    torch.manual_seed(0)
    gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
    enc = LiftSplatShoot(gc, dict(final_dim=(256, 512))).to(DEV).eval()       # eval for (1), (2): no drop-connect between two calls
    pts, masks = syn.robot_points_4()
    cfg = DPhysConfig(robot='tradr', grid_res=0.05, robot_points=pts, driving_parts=masks)
    dp = DPhysics(cfg, device=DEV)
    batch = synthetic_encoder_batch(enc, dp, n_rollouts=1024, device=DEV)
    inputs = batch[0]
    assert inputs[0].shape == (1, 4, 3, 256, 512)
    # (1) BEV of the fused lift-splat vs the oracle's voxel pooling of the lifted features
    with torch.no_grad():
        geom = enc.get_geometry(*inputs[1:])
        feats = enc.get_cam_feats(inputs[0])
        bev = enc.get_voxels(*inputs)
    assert bev.shape == (1, 64, 256, 256)
    ref, kept = so.voxel_pooling(geom.cpu().numpy(), feats.cpu().numpy(), enc.dx.cpu().numpy(), enc.bx.cpu().numpy(), enc.nx.cpu().numpy())
    assert hp.rel_err(bev.cpu(), ref) <= 1e-6
    # (2) the rollout of the step, on the maps the encoder predicts
    step = EncoderTrainStep(enc, dp, lr=1e-4)
    with torch.no_grad():
        terrain = enc(*inputs)
        zp = step.terrain_preproc(terrain['terrain']).squeeze(1)
        mp = step.terrain_preproc(terrain['friction']).squeeze(1)
        controls, pose0 = batch[3], batch[4]
        x0 = pose0[:, :3, 3].clone()
        st = (x0, torch.zeros_like(x0), pose0[:, :3, :3].contiguous(), torch.zeros_like(x0))
        (Xs, Xds, Rs, Om), _ = dp(z_grid=zp, controls=controls, state=st, friction=mp)
    assert Xs.shape == (1024, 500, 3) and zp.shape == (1, 256, 256)
    sub = torch.arange(0, 1024, 8)
    spec = hp.spec_from(pts, masks, 1, 0.05, 6.4)
    def ref_run(dtype):
        sr = tuple(t[sub].cpu().to(dtype) for t in (pose0[:, :3, 3], torch.zeros_like(x0), pose0[:, :3, :3], torch.zeros_like(x0)))
        with torch.no_grad():
            (a, b, c, d), _ = orc.rollout(spec, zp.cpu().to(dtype).expand(len(sub), -1, -1), controls[sub].cpu().to(dtype),
                                          friction=mp.cpu().to(dtype).expand(len(sub), -1, -1), state=sr)
        return [a, b, c, d]
    r32, r64 = ref_run(torch.float32), ref_run(torch.float64)
    n_calm = 0
    for o, a32, a64 in zip((Xs, Xds, Rs, Om), r32, r64):
        o = o[sub.to(DEV)].cpu().numpy().astype(np.float64)
        a32, a64 = a32.numpy().astype(np.float64), a64.numpy()
        Bq, T = a32.shape[:2]
        scale = np.abs(a64).reshape(Bq, -1).max(1).clip(1e-30)[:, None]
        env = np.maximum.accumulate(np.abs(a32 - a64).reshape(Bq, T, -1).max(2) / scale, axis=1)
        err = np.abs(o - a32).reshape(Bq, T, -1).max(2) / scale
        calm = env <= 1e-6
        n_calm += int(calm.sum())
        assert (err[calm] <= 1e-4).all(), float(err[calm].max())
        assert np.isfinite(o).all() and float(err.max()) < 0.5
    assert n_calm >= 4 * 128 * 10      # (a random-init encoder predicts a rough map: the reference itself stays reproducible for ~16 steps)
    # (3) the whole step: finite losses; gradients through rollout backward + lift-splat backward, looked at BEFORE the step's
    # clip_grad_norm_ (500-step BPTT on the rough map of a random-init encoder explodes -- the gradient norm overflows float32
    # and the clip then scales everything to zero, exactly as it would in the reference's train.py:167)
    enc.train()
    step.buckets.zero()
    parts = step.losses(batch)
    sum(parts).backward()
    step.buckets.finish()
    assert all(np.isfinite(float(v)) for v in parts)
    grads = [p.grad for p in enc.parameters() if p.requires_grad and p.grad is not None]
    assert len(grads) > 100
    dn = {n: (None if p.grad is None else float(p.grad.abs().max())) for n, p in enc.named_parameters() if n.startswith('camencode.depthnet')}
    assert any(v is not None and v > 0 for v in dn.values()), dn
    assert not any(bool(torch.isnan(t).any()) for t in grads)
    loss, parts2 = step.step(batch)          # and the full step (clip + Adam) runs
    assert np.isfinite(float(loss))


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('friction', [False, True])
@pytest.mark.parametrize('given_state', [False, True])
def test_component_parallel_mapping_vs_reference_golden(integ, friction, given_state):
    """The 16-lanes-per-rollout kernels (MF_LANES_COMPONENT; what small batches of a 4-point body run on by default) against
    the reference's own outputs and autograd gradients: case A of rollout_small.npz (N = 4, T = 48), plus oracle-checked
    variants with a friction map and a given start state (per-rollout maps, start poses off the origin)."""
    from tests.golden_state import given_state as make_state
    g = hp.load('rollout_small')
    pts, masks, z, ctrl, _, _ = hp.small_case(g, 'A', torch.float32)
    B = z.shape[0]
    mu = None
    if friction:
        from monoforce_amd import synthetic as syn
        mu = torch.stack([syn.wave_friction(hp.SMALL['d_max'], hp.SMALL['grid_res'], 0.5, 1.0, 1.1 + 0.2 * k, 0.8) for k in range(B)])
    state = tuple(t.float() for t in make_state(B)) if given_state else None
    dp = make_dphysics(pts, masks, integ, hp.SMALL['grid_res'], hp.SMALL['d_max'], points_per_lane=16)
    zd = z.to(DEV).requires_grad_(True)
    cd = ctrl.to(DEV).requires_grad_(True)
    md = None if mu is None else mu.to(DEV).requires_grad_(True)
    st = None if state is None else tuple(s.clone().to(DEV) for s in state)
    states, forces = dp(z_grid=zd, controls=cd, state=st, friction=md)
    outs = list(states) + list(forces)
    loss = hp.probe_loss(outs, torch.float32)
    loss.backward()
    if not friction and not given_state:       # the reference itself
        pre = f'A/f32/i{integ}/'
        for k, o in zip(hp.OUT_KEYS, outs):
            assert hp.rel_err(o, g[pre + k]) <= 1e-4, (k, hp.rel_err(o, g[pre + k]))
        assert abs(float(loss) - float(g[pre + 'loss'])) <= 2e-4 * abs(float(g[pre + 'loss'])) + 2e-4
        assert hp.rel_err(zd.grad, g[pre + 'g_z']) <= 2e-4, hp.rel_err(zd.grad, g[pre + 'g_z'])
        assert hp.rel_err(cd.grad, g[pre + 'g_ctrl']) <= 2e-4, hp.rel_err(cd.grad, g[pre + 'g_ctrl'])
    # ... and the oracle on the same inputs (float64 oracle: the float32 kernels are judged against exact arithmetic)
    spec = hp.spec_from(pts, masks, integ, hp.SMALL['grid_res'], hp.SMALL['d_max'])
    zc, cc = z.double().requires_grad_(True), ctrl.double().requires_grad_(True)
    mc = None if mu is None else mu.double().requires_grad_(True)
    sc = None if state is None else tuple(s.double().clone() for s in state)
    rs, rf = orc.rollout(spec, zc, cc, state=sc, friction=mc)
    refs = list(rs) + list(rf)
    hp.probe_loss(refs, torch.float64).backward()
    for k, o, r in zip(hp.OUT_KEYS, outs, refs):
        assert hp.rel_err(o, r) <= 1e-4, (k, hp.rel_err(o, r))
    assert hp.rel_err(zd.grad, zc.grad) <= 2e-4, hp.rel_err(zd.grad, zc.grad)
    assert hp.rel_err(cd.grad, cc.grad) <= 2e-4, hp.rel_err(cd.grad, cc.grad)
    if friction:
        assert hp.rel_err(md.grad, mc.grad) <= 2e-4, hp.rel_err(md.grad, mc.grad)
    if st is not None:
        assert hp.rel_err(st[0].cpu(), sc[0]) <= 1e-5       # the in-place terrain snap of the caller's start position


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('N', [1, 2, 3])
def test_component_parallel_mapping_with_fewer_points(N, integ):
    """Bodies of fewer than 4 contact points leave quads of the 16-lane row without a point: they must contribute nothing."""
    from monoforce_amd import synthetic as syn
    pts4, _ = syn.robot_points_4()
    pts = pts4[:N].copy()
    if N < 3:
        pts[:, 2] += np.array([0.0, 0.05])[:N]        # (inertia of 1 or 2 points is singular: the test supplies its own)
    masks = [pts[:, 1] > 0, pts[:, 1] <= 0]
    B, T = 5, 40
    z = torch.stack([syn.bump_terrain(syn.bump_params(20 + k), 1.6, 0.1) * 0.3 for k in range(B)])
    ctrl = syn.varying_controls(B, T, seed=3)
    base = make_dphysics(pts4, [pts4[:, 1] > 0, pts4[:, 1] <= 0], integ, 0.1, 1.6)
    outs = {}
    for ppl in (16, 1):
        dp = make_dphysics(pts4, [pts4[:, 1] > 0, pts4[:, 1] <= 0], integ, 0.1, 1.6, points_per_lane=ppl)
        dp.dphys_cfg.robot_points = torch.as_tensor(pts)
        dp.dphys_cfg.driving_parts = [torch.as_tensor(m) for m in masks]
        dp.x_points = dp.dphys_cfg.robot_points.unsqueeze(0).to(dp.device)
        dp._cache = {('iinv', torch.float32): base._iinv(torch.float32)}      # the 4-point body's inertia for both mappings
        zd = z.to(DEV).requires_grad_(True)
        st, fo = dp(zd, ctrl.to(DEV))
        (st[0].square().sum() + 1e-3 * fo[0].square().sum()).backward()
        outs[ppl] = [o.detach().cpu() for o in list(st) + list(fo)] + [zd.grad.cpu()]
    for k, a_, b_ in zip(hp.OUT_KEYS + ('g_z',), outs[16], outs[1]):
        assert a_.shape == b_.shape and torch.isfinite(a_).all()
        assert hp.rel_err(a_, b_) <= (2e-4 if k == 'g_z' else 1e-4), (k, hp.rel_err(a_, b_))


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('B', [5, 1024])
def test_component_parallel_backward_positions_only_loss(B, integ):
    """A loss that reads the positions only (what `physics_loss` does, losses.py:102-127) takes the backward kernel with the
    other five upstream gradients compiled out: its gradients vs the oracle (B = 5 per-rollout maps, every rollout in the loss;
    B = 1024 on one shared map, 32 rollouts in the loss)."""
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    T = 80
    shared = B > 16
    nz = 1 if shared else B
    z = torch.stack([syn.bump_terrain(syn.bump_params(50 + k), 6.4, 0.05) * 0.7 for k in range(nz)])
    mu = torch.stack([syn.wave_friction(6.4, 0.05, 0.5, 1.0, 1.2 + 0.1 * k, 0.9) for k in range(nz)])
    ctrl = syn.varying_controls(B, T, seed=6)
    sel = torch.arange(0, B, max(B // 32, 1))[:32]
    wts = syn.probe_weights((len(sel), T, 3), phase=0.7)
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4, points_per_lane=16)
    zd, md, cd = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True), ctrl.to(DEV).requires_grad_(True)
    (Xs, _, _, _), _ = dp(zd, cd, friction=md)
    (Xs[sel.to(DEV)] * wts.to(DEV)).sum().backward()
    spec = hp.spec_from(pts, masks, integ, 0.05, 6.4)
    zc, mc = z.double().requires_grad_(True), mu.double().requires_grad_(True)
    cc = ctrl[sel].double().requires_grad_(True)
    zin = zc.expand(len(sel), -1, -1) if shared else zc[sel]
    min_ = mc.expand(len(sel), -1, -1) if shared else mc[sel]
    (rX, _, _, _), _ = orc.rollout(spec, zin, cc, friction=min_)
    (rX * wts.double()).sum().backward()
    assert hp.rel_err(zd.grad, zc.grad) <= 2e-4, hp.rel_err(zd.grad, zc.grad)
    assert hp.rel_err(md.grad, mc.grad) <= 2e-4, hp.rel_err(md.grad, mc.grad)
    assert hp.rel_err(cd.grad[sel.to(DEV)], cc.grad) <= 2e-4, hp.rel_err(cd.grad[sel.to(DEV)], cc.grad)


@pytest.mark.parametrize('ppl', [0, 1, 16])
def test_dynamics_rodrigues_step_of_a_fast_spinning_body(ppl):
    """`dynamics()`'s rotation update, R <- R (I + K sin(|w| dt) + K^2 (1 - cos(|w| dt))) (dphysics.py:274-288), away from the
    |w| dt ~ 1e-2 of a driving robot: start spins of 20 .. 400 rad/s cover the series (|w| dt < 1) and the hardware sin / cos of
    the fast-math kernels beyond it, forward and backward, against the float64 oracle."""
    from tests.golden_state import given_state as make_state
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    B, T = 6, 6
    z = torch.stack([syn.bump_terrain(syn.bump_params(70 + k), 1.6, 0.1) * 0.2 for k in range(B)])
    ctrl = syn.varying_controls(B, T, seed=11)
    x, xd, R, w = make_state(B)
    spin = torch.tensor([20.0, 60.0, 99.0, 101.0, 250.0, 400.0], dtype=torch.float64)
    w = torch.stack([0.3 * spin, -0.2 * spin, 0.93 * spin], 1)
    state = (x, xd, R, w)
    dp = make_dphysics(pts, masks, 0, 0.1, 1.6, points_per_lane=ppl)
    zd = z.to(DEV).requires_grad_(True)
    st = tuple(s.float().clone().to(DEV) for s in state)
    (Xs, Xds, Rs, Om), _ = dp(z_grid=zd, controls=ctrl.to(DEV), state=st)
    wts = syn.probe_weights((B, T, 3, 3), phase=0.3)
    ((Rs * wts.to(DEV)).sum() + Xs.square().sum()).backward()
    spec = hp.spec_from(pts, masks, 0, 0.1, 1.6)
    zc = z.double().requires_grad_(True)
    (rX, rXd, rR, rOm), _ = orc.rollout(spec, zc, ctrl.double(), state=tuple(s.clone() for s in state))
    ((rR * wts.double()).sum() + rX.square().sum()).backward()
    for b in range(B):
        assert hp.rel_err(Rs[b], rR[b]) <= 2e-5, (b, hp.rel_err(Rs[b], rR[b]))
        assert hp.rel_err(Xs[b], rX[b]) <= 1e-4, (b, hp.rel_err(Xs[b], rX[b]))
    assert hp.rel_err(zd.grad, zc.grad) <= 2e-4, hp.rel_err(zd.grad, zc.grad)


_NO_RECORD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from tests.test_parity_net_gpu import _record_case
torch.save(_record_case(%d, %d), %r)
'''


def _record_case(integ=1, B=37):
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    T = 90
    z = syn.bump_terrain(syn.bump_params(61), 6.4, 0.05) * 0.8
    mu = syn.wave_friction(6.4, 0.05, 0.5, 1.0, 1.3, 0.9)
    ctrl = syn.varying_controls(B, T, seed=9)
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4)
    zd, md, cd = z.to(DEV).requires_grad_(True), mu.to(DEV).requires_grad_(True), ctrl.to(DEV).requires_grad_(True)
    st, fo = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0))
    outs = list(st) + list(fo)
    hp.probe_loss(outs, torch.float32).backward()
    keep = slice(0, 64)       # (the larger batches: outputs of the first rollouts only -- the gradients carry all of them)
    return dict(outs=[o.detach()[keep].cpu() for o in outs], gz=zd.grad.cpu(), gmu=md.grad.cpu(), gc=cd.grad.cpu())


# B = 37: one workgroup per CU (twelve-slot ring); 1500 rollouts = 375 workgroups: two per CU (six-slot ring); 3000 rollouts = 750
# waves: beyond the streaming form, the record read by one wave.  dynamics() (integ 0) streams up to 256 workgroups (B = 37) and keeps no
# record beyond (read by one wave it loses to recomputing): MF_CP_RECORD_DYNAMICS=1 puts that kernel through the same comparison (1500).
@pytest.mark.parametrize('integ,B', [(1, 37), (0, 37), (1, 1500), (1, 3000), (0, 1500)])
def test_backward_from_the_forward_record_equals_the_recomputing_backward(integ, B):
    """Launches of up to one wave per SIMD keep a compact per-step record in the forward (MfRolloutFwdBufs.rec, 256 B per
    rollout-step) and the backward reads it instead of recomputing (default integrator: through a second wave that streams it
    into LDS); a child process with MF_CP_RECORD_MAX_WAVES=0 runs the same problem through the recomputing backward: same outputs
    bit for bit (the record changes no arithmetic of the forward), gradients to float32 rounding, and -- at the small batch --
    both within the usual bar of the float64 oracle."""
    import os, subprocess, sys, tempfile
    from oracle import dphysics_oracle as orc  # noqa: F401
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        if integ == 0:       # (the library reads its switches once per process: the recorded run of dynamics() is a child as well)
            path0 = os.path.join(td, 'rec.pt')
            r = subprocess.run([sys.executable, '-c', _NO_RECORD % (repo, integ, B, path0)], capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, MF_CP_RECORD_DYNAMICS='1'))
            assert r.returncode == 0, r.stderr[-2000:]
            with_rec = torch.load(path0)
        else:
            with_rec = _record_case(integ, B)
        path = os.path.join(td, 'norec.pt')
        env = dict(os.environ, MF_CP_RECORD_MAX_WAVES='0')
        r = subprocess.run([sys.executable, '-c', _NO_RECORD % (repo, integ, B, path)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        without = torch.load(path)
        # the record read by the computing wave itself (MF_CP_BWD_MODE=2) instead of streamed through LDS by a second wave
        # (the default integrator's default): the same arithmetic, the adjoint state summed over the contact points once instead of every step
        path1 = os.path.join(td, 'onewave.pt')
        r = subprocess.run([sys.executable, '-c', _NO_RECORD % (repo, integ, B, path1)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MF_CP_BWD_MODE='2', MF_CP_RECORD_DYNAMICS='1'))
        assert r.returncode == 0, r.stderr[-2000:]
        one_wave = torch.load(path1)
    for a_, b_ in zip(with_rec['outs'], without['outs']):
        assert torch.equal(a_, b_)
    tol = 2e-5 if B <= 64 else 1e-4       # (thousands of rollouts add up in the map gradients: float32 sums in another order)
    for k in ('gz', 'gmu', 'gc'):
        assert hp.rel_err(with_rec[k], without[k]) <= tol, (k, hp.rel_err(with_rec[k], without[k]))
        assert hp.rel_err(with_rec[k], one_wave[k]) <= tol, (k, hp.rel_err(with_rec[k], one_wave[k]))      # (sums in another order)
    if B > 64:
        return
    # ... and the recorded route against the ORACLE (both integrators).  Default integrator: the float32 kernel against the float64
    # oracle.  dynamics(): over these 90 steps of rough terrain its float32 and float64 runs part ways (SURVEY fact 6 -- the float32
    # ORACLE does too), so its referee is the float64 build of the same kernels (points_per_lane = 16 on float64 inputs: the record
    # read by the computing wave) against the float64 oracle, at 1e-7 -- not another HIP kernel.
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    T = 90
    z = (syn.bump_terrain(syn.bump_params(61), 6.4, 0.05) * 0.8).double().requires_grad_(True)
    mu = syn.wave_friction(6.4, 0.05, 0.5, 1.0, 1.3, 0.9).double().requires_grad_(True)
    ctrl = syn.varying_controls(B, T, seed=9).double().requires_grad_(True)
    spec = hp.spec_from(pts, masks, integ, 0.05, 6.4)
    rs, rf = orc.rollout(spec, z.unsqueeze(0).expand(B, -1, -1), ctrl, friction=mu.unsqueeze(0).expand(B, -1, -1))
    hp.probe_loss(list(rs) + list(rf), torch.float64).backward()
    if integ == 1:
        bar = 2e-4
        assert hp.rel_err(with_rec['gz'], z.grad) <= bar and hp.rel_err(with_rec['gmu'], mu.grad) <= bar and hp.rel_err(with_rec['gc'], ctrl.grad) <= bar
        return
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4, points_per_lane=16)
    zd, md, cd = (t.detach().to(DEV).requires_grad_(True) for t in (z, mu, ctrl))
    st, fo = dp(zd.unsqueeze(0), cd, friction=md.unsqueeze(0))
    hp.probe_loss(list(st) + list(fo), torch.float64).backward()
    for k, a_, b_ in (('gz', zd.grad, z.grad), ('gmu', md.grad, mu.grad), ('gc', cd.grad, ctrl.grad)):
        assert hp.rel_err(a_, b_) <= 1e-7, (k, hp.rel_err(a_, b_))


@pytest.mark.parametrize('integ', [1, 0])
@pytest.mark.parametrize('loss_on', ['all', 'positions'])
def test_streaming_backward_every_horizon_remainder(loss_on, integ):
    """The backward that streams the forward's record through LDS (rollout_bwd_cp_kernel.h, MODE = kCpStream) fetches in batches
    of three steps through a three-stage pipeline (loads | cell gathers | rebuild) over three register sets, drains it in one
    of three ways and peels the last steps; the computing wave runs two steps per trip (dynamics(): one wave reads the record,
    two steps per trip).  Horizons 1 .. 26 and 47 / 48 put every prologue, drain and remainder of those loops through it
    (all-outputs loss with control gradients, and the positions-only variant of the training loss), against the
    one-point-per-lane kernels: float32 rounding apart."""
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    B = 6
    z = torch.stack([syn.bump_terrain(syn.bump_params(20 + k), 1.6, 0.1) * 0.3 for k in range(B)])
    mu = torch.stack([syn.wave_friction(1.6, 0.1, 0.5, 1.0, 1.1 + 0.2 * k, 0.8) for k in range(B)])
    for T in list(range(1, 27)) + [47, 48]:
        ctrl = syn.varying_controls(B, max(T, 2), seed=3)[:, :T]
        res = {}
        for ppl in (16, 1):
            dp = make_dphysics(pts, masks, integ, 0.1, 1.6, points_per_lane=ppl)
            zd, md, cd = z.cuda().requires_grad_(True), mu.cuda().requires_grad_(True), ctrl.cuda().requires_grad_(True)
            st, fo = dp(zd, cd, friction=md)
            if loss_on == 'all':
                hp.probe_loss(list(st) + list(fo), torch.float32).backward()
            else:
                (st[0] * syn.probe_weights(st[0].shape, phase=0.4).cuda()).sum().backward()
            res[ppl] = (zd.grad.cpu(), md.grad.cpu(), cd.grad.cpu())
        for name, a_, b_ in zip(('gz', 'gmu', 'gctrl'), res[16], res[1]):
            if float(b_.abs().max()) == 0.0:
                assert float(a_.abs().max()) == 0.0, (T, name)
            else:
                assert hp.rel_err(a_, b_) <= 2e-5, (T, name, hp.rel_err(a_, b_))
