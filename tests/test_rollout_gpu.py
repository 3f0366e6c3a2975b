"""Parity of the HIP rollout (through the C ABI) against the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from tests import helpers as hp

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def make_dphysics(points, masks, integ, grid_res, d_max, **kw):
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    n_tracks = len(masks)
    cfg = DPhysConfig(robot='tradr' if n_tracks == 2 else 'husky', grid_res=grid_res, robot_points=points, driving_parts=masks)
    cfg.robot_mass = 40.0
    cfg.damping = float(np.sqrt(4 * cfg.robot_mass * cfg.stiffness))
    cfg.d_max = d_max
    cfg.use_odeint = (integ == 1)
    return DPhysics(cfg, device=DEV, **kw)


def run_hip(dp, z, ctrl, state, mu):
    st = None if state is None else tuple(s.clone().to(DEV) for s in state)
    states, forces = dp(z_grid=z.to(DEV), controls=ctrl.to(DEV), state=st, friction=None if mu is None else mu.to(DEV))
    torch.cuda.synchronize()
    return [o.cpu() for o in list(states) + list(forces)], st


@pytest.mark.parametrize('name', ['A', 'B', 'C'])
@pytest.mark.parametrize('tag', ['f32', 'f64'])
@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('ppl', [1, 4])
@pytest.mark.parametrize('precise', [False, True])
def test_small_rollout_vs_reference_golden(name, tag, integ, ppl, precise):
    """B<=3, T=48, 32x32: all six outputs of the HIP path vs the reference's own outputs."""
    g = hp.load('rollout_small')
    dt = hp.DT[tag]
    pts, masks, z, ctrl, state, mu = hp.small_case(g, name, dt)
    dp = make_dphysics(pts, masks, integ, hp.SMALL['grid_res'], hp.SMALL['d_max'], points_per_lane=ppl, precise=precise)
    outs, st = run_hip(dp, z, ctrl, state, mu)
    # float64: agreement to rounding.  float32: north_star's bar, <= 1e-4 rel on poses and forces.
    tol = 1e-9 if tag == 'f64' else 1e-4
    for k, o in zip(hp.OUT_KEYS, outs):
        ref = g[f'{name}/{tag}/i{integ}/{k}']
        assert tuple(o.shape) == ref.shape
        assert hp.rel_err(o, ref) <= tol, (k, hp.rel_err(o, ref))
    if st is not None:      # the in-place terrain snap of the caller's x (dphysics.py:571)
        assert hp.rel_err(st[0].cpu(), g[f'{name}/{tag}/i{integ}/x0_after']) <= tol


@pytest.mark.parametrize('tag', ['f32', 'f64'])
@pytest.mark.parametrize('ppl', [1, 4])
@pytest.mark.parametrize('precise', [False, True])
def test_teacher_forced_single_step(tag, ppl, precise):
    """One step from the reference's own mid-rollout states: state -> forces, next state (<= 1e-5 rel in fp32)."""
    g = hp.load('step'); gs = hp.load('rollout_small')
    dt = hp.DT[tag]
    pts, masks, z, ctrl, _, mu = hp.small_case(gs, 'B', dt)
    dp = make_dphysics(pts, masks, 0, hp.SMALL['grid_res'], hp.SMALL['d_max'], snap_to_terrain=False, points_per_lane=ppl, precise=precise)
    tol = 1e-11 if tag == 'f64' else 1e-5
    for t in g['sel']:
        st = tuple(torch.as_tensor(g[f'{tag}/t{t}/in_{k}']) for k in ('x', 'xd', 'R', 'w'))
        outs, _ = run_hip(dp, z, ctrl[:, t + 1:t + 2], st, mu)
        sink = 40.0 * 9.81 / (50_000. + 1e-6)
        nxt = {k: g[f'{tag}/t{t}/next_{k}'] for k in ('x', 'xd', 'R', 'w')}
        assert hp.rel_err(outs[0][:, 0], nxt['x'] + nxt['R'][:, :, 2] * sink) <= tol
        assert hp.rel_err(outs[1][:, 0], nxt['xd']) <= tol
        assert hp.rel_err(outs[2][:, 0], nxt['R']) <= tol
        assert hp.rel_err(outs[3][:, 0], nxt['w']) <= tol
        assert hp.rel_err(outs[4][:, 0], g[f'{tag}/t{t}/Fs']) <= tol
        assert hp.rel_err(outs[5][:, 0], g[f'{tag}/t{t}/Ff']) <= tol


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('ppl', [1, 4, 0, 16])      # 16: the float64 build of the component-parallel kernels every BASELINE config runs on
def test_full_horizon_f64_vs_reference(integ, ppl):
    """T=500 on 256x256 in float64: chaos-proof full-horizon parity with the reference (<= 1e-8 rel)."""
    g = hp.load('rollout_full')
    pts, masks, z, mu, ctrl = hp.full_inputs(torch.float64)
    dp = make_dphysics(pts, masks, integ, hp.FULL['grid_res'], hp.FULL['d_max'], points_per_lane=ppl)
    outs, _ = run_hip(dp, z, ctrl, None, mu)
    for k, o in zip(hp.OUT_KEYS[:4], outs[:4]):
        assert hp.rel_err(o, g[f'f64/i{integ}/{k}']) <= 1e-8, k
    assert hp.rel_err(outs[4][:, ::10], g[f'f64/i{integ}/Fs_10']) <= 1e-7
    assert hp.rel_err(outs[5][:, ::10], g[f'f64/i{integ}/Ff_10']) <= 1e-7


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('ppl,precise', [(1, False), (1, True), (4, False), (4, True), (0, False), (16, False)])   # 0 / 16: the component-parallel kernels every BASELINE config runs on
def test_full_horizon_f32_within_reference_envelope(integ, ppl, precise):
    """T=500 free-run in float32.  The rollout is chaotic: the reference's own fp32 and fp64 runs drift
    apart exponentially, and so does any other float32 evaluation order (e.g. the two lane mappings of the kernel).
    Bar: north_star's <= 1e-4 rel on every step up to which the reference itself is reproducible across precisions
    (its own running fp32-vs-fp64 envelope <= 1e-6); beyond that only boundedness is asserted.  The non-chaotic
    rollouts (default integrator on smooth / flat terrain) must meet 1e-4 over the whole 500-step horizon."""
    g = hp.load('rollout_full')
    pts, masks, z, mu, ctrl = hp.full_inputs(torch.float32)
    dp = make_dphysics(pts, masks, integ, hp.FULL['grid_res'], hp.FULL['d_max'], points_per_lane=ppl, precise=precise)
    outs, _ = run_hip(dp, z, ctrl, None, mu)
    n_calm = 0
    for k, o in zip(hp.OUT_KEYS[:4], outs[:4]):
        r32, r64 = g[f'f32/i{integ}/{k}'].astype(np.float64), g[f'f64/i{integ}/{k}']
        o = o.numpy().astype(np.float64)
        B, T = r32.shape[:2]
        scale = np.abs(r64).reshape(B, -1).max(1).clip(1e-30)[:, None]
        env = np.abs(r32 - r64).reshape(B, T, -1).max(2) / scale          # reference fp32 vs fp64, per rollout and step
        err = np.abs(o - r32).reshape(B, T, -1).max(2) / scale            # ours vs reference fp32
        env_run = np.maximum.accumulate(env, axis=1)
        calm = env_run <= 1e-6
        n_calm += int(calm.sum())
        assert (err[calm] <= 1e-4).all(), (k, float(err[calm].max()))
        assert np.isfinite(o).all() and float(err.max()) < 0.5, (k, float(err.max()))
    assert n_calm > 4 * 4 * 60
    if integ == 1:
        # the reference's default integrator on the smooth / flat terrains (rollouts 2, 3) is not chaotic:
        # there the whole 500-step horizon meets north_star's 1e-4
        for k, o in zip(hp.OUT_KEYS[:4], outs[:4]):
            assert hp.rel_err(o[2:], g[f'f32/i1/{k}'][2:]) <= 1e-4, k


@pytest.mark.parametrize('N,n_tracks', [(3, 2), (7, 2), (16, 2), (33, 4), (64, 2), (100, 4), (175, 2), (223, 4), (300, 2)])
@pytest.mark.parametrize('ppl', [1, 4])
def test_point_counts_vs_oracle_f64(N, n_tracks, ppl):
    """Every lane-group / points-per-lane instantiation (N = 175 tradr, 223 marv in the reference) vs the CPU oracle."""
    from monoforce_amd import synthetic as syn
    from oracle import dphysics_oracle as orc
    pts, masks = syn.robot_points_box(N, seed=N, n_tracks=n_tracks)
    B, T = 3, 40
    z = torch.stack([syn.bump_terrain(syn.bump_params(20 + b), 3.2, 0.1, torch.float64) * 0.3 for b in range(B)])
    mu = torch.stack([syn.wave_friction(3.2, 0.1, 0.5, 1.0, 1.0 + b, 0.8, torch.float64) for b in range(B)])
    ctrl = syn.varying_controls(B, T, seed=N, dtype=torch.float64)
    for integ in (0, 1):
        spec = hp.spec_from(pts, masks, integ, 0.1, 3.2)
        with torch.no_grad():
            st, fo = orc.rollout(spec, z, ctrl, friction=mu)
        dp = make_dphysics(pts, masks, integ, 0.1, 3.2, points_per_lane=ppl)
        outs, _ = run_hip(dp, z, ctrl, None, mu)
        for k, o, r in zip(hp.OUT_KEYS, outs, list(st) + list(fo)):
            assert hp.rel_err(o, r) <= 1e-9, (N, integ, k, hp.rel_err(o, r))


@pytest.mark.parametrize('N,n_tracks', [(100, 4), (175, 2), (223, 4), (300, 2)])
@pytest.mark.parametrize('integ', [0, 1])
def test_large_bodies_fast_f32_multi_wave_kernels(N, n_tracks, integ):
    """The float32 fast-math kernels that spread ONE rollout over several waves (G = 128 / 256 / 512 lanes, one point per lane -- the
    mapping the reference's own robots, 175 / 223 points, run on at small batches; the default integrator's are software-pipelined
    around their two LDS exchanges, rollout_fwd_kernel.h PIPE): against the float64 oracle on the same inputs, against the
    one-wave mapping with 2 / 4 / 8 points per lane (another kernel: sums in another order, no pipelining), with forces and states only."""
    from monoforce_amd import synthetic as syn
    from oracle import dphysics_oracle as orc
    pts, masks = syn.robot_points_box(N, seed=N, n_tracks=n_tracks)
    B, T = 5, 60
    z64 = torch.stack([syn.bump_terrain(syn.bump_params(40 + b), 3.2, 0.1, torch.float64) * 0.3 for b in range(B)])
    mu64 = torch.stack([syn.wave_friction(3.2, 0.1, 0.5, 1.0, 1.0 + b, 0.8, torch.float64) for b in range(B)])
    ctrl64 = syn.varying_controls(B, T, seed=N, dtype=torch.float64)
    spec = hp.spec_from(pts, masks, integ, 0.1, 3.2)
    with torch.no_grad():
        st, fo = orc.rollout(spec, z64, ctrl64, friction=mu64)
    ref = list(st) + list(fo)
    z, mu, ctrl = z64.float(), mu64.float(), ctrl64.float()
    wide, _ = run_hip(make_dphysics(pts, masks, integ, 0.1, 3.2), z, ctrl, None, mu)                       # auto: G > 64 at this batch size
    narrow, _ = run_hip(make_dphysics(pts, masks, integ, 0.1, 3.2, points_per_lane=4), z, ctrl, None, mu)   # one wave, several points per lane
    for k, a, b, r in zip(hp.OUT_KEYS, wide, narrow, ref):
        # float32 vs float64 over 60 steps of a stiff contact model; dynamics() reports the step's own forces (not impulses summed
        # over the horizon), which carry the state's rounding through the stiffness
        tol = (1e-2 if integ == 0 else 2e-3) if k in ('Fs', 'Ff') else 2e-4
        assert hp.rel_err(a, r.float()) <= tol, (N, integ, k, 'vs oracle', hp.rel_err(a, r.float()))
        assert hp.rel_err(a, b) <= 5e-5, (N, integ, k, 'vs one-wave mapping', hp.rel_err(a, b))
    dp = make_dphysics(pts, masks, integ, 0.1, 3.2)
    dp.return_forces = False                                                                              # the states-only kernels
    lean, forces = dp(z_grid=z.to(DEV), controls=ctrl.to(DEV), friction=mu.to(DEV))
    assert forces == (None, None)
    for k, a, b in zip(hp.OUT_KEYS[:4], lean, wide[:4]):
        assert torch.equal(a.cpu(), b), (N, integ, k)


def _c2_inputs(B, T=500, dtype=torch.float32):
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    z = syn.bump_terrain(syn.bump_params(0), 6.4, 0.05, dtype)
    mu = syn.wave_friction(6.4, 0.05, dtype=dtype)
    ctrl = syn.const_controls(B, T, seed=0, dtype=dtype)
    return pts, masks, z, mu, ctrl


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('B', [1024, 256])      # BASELINE configs[2] / configs[1] (C2: 256 rollouts, forward) at their own sizes
def test_full_size_properties(integ, B):
    """BASELINE config sizes (B=1024 / 256, T=500, N=4, 256x256): size-independent properties.
    determinism; shared (expanded) map == per-rollout copies; time-major == batch-major; workgroup size irrelevant;
    rollouts independent of their batch neighbours; finite outputs."""
    pts, masks, z, mu, ctrl = _c2_inputs(B)
    zs, ms = z.to(DEV).unsqueeze(0).expand(B, -1, -1), mu.to(DEV).unsqueeze(0).expand(B, -1, -1)
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4)
    ref, _ = run_hip(dp, zs, ctrl, None, ms)
    assert all(torch.isfinite(o).all() for o in ref)
    again, _ = run_hip(dp, zs, ctrl, None, ms)
    per_rollout, _ = run_hip(dp, zs.contiguous(), ctrl, None, ms.contiguous())
    bm, _ = run_hip(make_dphysics(pts, masks, integ, 0.05, 6.4, contiguous_outputs=True), zs, ctrl, None, ms)
    wg256, _ = run_hip(make_dphysics(pts, masks, integ, 0.05, 6.4, block=256), zs, ctrl, None, ms)
    packed, _ = run_hip(make_dphysics(pts, masks, integ, 0.05, 6.4, points_per_lane=4), zs, ctrl, None, ms)
    packed2, _ = run_hip(make_dphysics(pts, masks, integ, 0.05, 6.4, points_per_lane=4, block=256), zs, ctrl, None, ms)
    for a, b in zip(packed, packed2):
        assert torch.equal(a, b)
    # the two lane mappings differ only in summation order: identical for the first steps, close while not chaotic
    assert hp.rel_err(packed[0][:, :50], ref[0][:, :50]) <= 1e-5
    sub, _ = run_hip(dp, zs[:100], ctrl[37:137], None, ms[:100])
    for k, a, b, c, d, e, f in zip(hp.OUT_KEYS, ref, again, per_rollout, bm, wg256, sub):
        assert torch.equal(a, b), f'{k}: not deterministic'
        assert torch.equal(a, c), f'{k}: shared map != per-rollout map'
        assert torch.equal(a, d), f'{k}: layouts differ'
        assert d.is_contiguous() and not a.is_contiguous()
        assert torch.equal(a, e), f'{k}: workgroup size changes results'
        assert torch.equal(a[37:137], f), f'{k}: rollouts are not independent'
    # physical sanity: the robot stays on the map and moves
    X = ref[0]
    assert float(X[..., :2].abs().max()) < 6.4 and float((X[:, -1, :2] - X[:, 0, :2]).norm(dim=-1).mean()) > 0.3


def test_edge_cases_and_errors():
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    z = torch.zeros(1, 16, 16)
    dp = make_dphysics(pts, masks, 1, 0.1, 0.8)
    # T = 1 with the default integrator: only the initial state comes back
    (Xs, Xds, Rs, Om), (Fs, Ff) = dp(z.to(DEV), torch.tensor([[[0.5, 0.1]]]).to(DEV))
    assert Xs.shape == (1, 1, 3) and Fs.shape == (1, 1, 4, 3)
    assert torch.equal(Rs[0, 0].cpu(), torch.eye(3)) and float(Fs.abs().max()) == 0.0
    assert abs(float(Xds[0, 0, 0]) - 0.5) < 1e-7 and abs(float(Om[0, 0, 2]) - 0.1) < 1e-7
    # shape assert with the reference's message (dphysics.py:575)
    dp2 = make_dphysics(pts, masks, 1, 0.1, 0.8)
    with pytest.raises(AssertionError, match='Controls shape'):
        dp2(torch.zeros(2, 16, 16).to(DEV), torch.zeros(2, 600, 2).to(DEV))      # longer than int(T/dt) = 500
    # self.ts is truncated for good, as in the reference (dphysics.py:581): a longer horizon afterwards trips the assert
    dp2(torch.zeros(2, 16, 16).to(DEV), torch.zeros(2, 10, 2).to(DEV))
    assert dp2.ts.shape[0] == 10
    # CPU tensors are refused -- there is no CPU fallback
    dp3 = make_dphysics(pts, masks, 1, 0.1, 0.8)
    dp3.device = 'cpu'
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dp3(torch.zeros(1, 16, 16), torch.zeros(1, 10, 2))
    # friction=None equals an all-ones friction map (cfg.friction, dphysics.py:562)
    ctrl = syn.const_controls(4, 60, seed=3).to(DEV)
    zz = syn.bump_terrain(syn.bump_params(3), 0.8, 0.1).to(DEV).unsqueeze(0).expand(4, -1, -1) * 0.2
    a = make_dphysics(pts, masks, 0, 0.1, 0.8)(zz, ctrl)
    b = make_dphysics(pts, masks, 0, 0.1, 0.8)(zz, ctrl, friction=torch.ones(4, 16, 16, device=DEV))
    for u, v in zip(a[0] + a[1], b[0] + b[1]):
        assert torch.equal(u, v)


@pytest.mark.parametrize('tag,precise', [('f32', False), ('f32', True), ('f64', False)])
@pytest.mark.parametrize('integ', [0, 1])
def test_flipper_joint_angles_vs_reference(tag, precise, integ):
    """robot == 'marv' with moving flippers (update_joints + per-step inertia, dphysics.py:192-197, 326-358) vs the reference."""
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    g = hp.load('rollout_joints')
    dt = hp.DT[tag]
    cfg = DPhysConfig(robot='marv', grid_res=0.1, robot_points=g['points'], driving_parts=g['masks'])
    cfg.robot_mass = 40.0
    cfg.damping = float(np.sqrt(4 * cfg.robot_mass * cfg.stiffness))
    cfg.d_max, cfg.use_odeint = 1.6, (integ == 1)
    assert np.allclose(np.array(list(cfg.joint_positions.values())), g['joint_positions'])
    dp = DPhysics(cfg, device=DEV, precise=precise)      # float32: fast-math and exact articulated kernels
    t = lambda k: torch.as_tensor(g[k]).to(dt).to(DEV)  # noqa: E731
    with torch.no_grad():
        states, forces = dp(t('z'), t('ctrl'), joint_angles=t('joint_angles'), friction=t('mu'))
    tol = 1e-9 if tag == 'f64' else 1e-4
    for k, o in zip(hp.OUT_KEYS, list(states) + list(forces)):
        assert hp.rel_err(o.cpu(), g[f'{tag}/i{integ}/{k}']) <= tol, (k, hp.rel_err(o.cpu(), g[f'{tag}/i{integ}/{k}']))
    # zero angles take the rigid-body kernels and agree with passing no angles at all (the reference's short-circuit, :340)
    a = dp(t('z'), t('ctrl'), joint_angles=torch.zeros_like(t('joint_angles')), friction=t('mu'))
    b = dp(t('z'), t('ctrl'), friction=t('mu'))
    assert all(torch.equal(u, v) for u, v in zip(a[0] + a[1], b[0] + b[1]))
    # gradients through the articulated rollout (joint angles are constants) vs the reference's loss.backward()
    from monoforce_amd import synthetic as syn
    zg, cg, mg = t('z').requires_grad_(True), t('ctrl').requires_grad_(True), t('mu').requires_grad_(True)
    jg = t('joint_angles').requires_grad_(True)          # ... and w.r.t. the angles: update_joints + the per-step inertia
    states, forces = dp(zg, cg, joint_angles=jg, friction=mg)
    loss = 0
    for i, (o, sc) in enumerate(zip(list(states) + list(forces), [1.0, 1.0, 1.0, 1.0, 1e-3, 1e-3])):
        loss = loss + (o * syn.probe_weights(o.shape, phase=0.5 + i, dtype=dt).to(DEV)).sum() * sc
    loss.backward()
    pre = f'{tag}/i{integ}/'
    gtol = 1e-8 if tag == 'f64' else 2e-4
    assert abs(float(loss) - float(g[pre + 'loss'])) <= gtol * abs(float(g[pre + 'loss'])) + gtol
    for k, v in (('g_z', zg.grad), ('g_ctrl', cg.grad), ('g_mu', mg.grad), ('g_ja', jg.grad)):
        assert hp.rel_err(v.cpu(), g[pre + k]) <= gtol, (k, hp.rel_err(v.cpu(), g[pre + k]))


def test_dtype_device_and_stride_handling():
    """Inputs that are non-contiguous, on the wrong device or of mixed dtype are normalised like `.to(device)` would."""
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    B, T = 5, 50
    dp = make_dphysics(pts, masks, 1, 0.1, 3.2)
    z = torch.stack([syn.bump_terrain(syn.bump_params(70 + b), 3.2, 0.1) * 0.3 for b in range(B)])
    ctrl = syn.varying_controls(B, T, seed=1)
    ref = dp(z.to(DEV), ctrl.to(DEV))
    # CPU inputs are moved; float64 controls are cast to the map's dtype; a transposed-storage map is made contiguous
    zt = z.transpose(1, 2).contiguous().transpose(1, 2)
    out = dp(zt, ctrl.double())
    for a, b in zip(ref[0] + ref[1], out[0] + out[1]):
        assert a.dtype == torch.float32 and torch.equal(a, b)
    # a given state whose x is a non-contiguous view still receives the terrain snap in place (dphysics.py:571)
    pose = torch.eye(4).repeat(B, 1, 1).to(DEV)
    x_view = pose[:, :3, 3]
    st = (x_view, torch.zeros(B, 3, device=DEV), pose[:, :3, :3], torch.zeros(B, 3, device=DEV))
    dp(z.to(DEV), ctrl.to(DEV), state=st)
    assert float(pose[:, 2, 3].abs().max()) > 0 and torch.equal(pose[:, 2, 3], x_view[:, 2])
    # half precision is refused loudly
    with pytest.raises(TypeError):
        dp(z.half().to(DEV), ctrl.to(DEV))


@pytest.mark.parametrize('integ', [0, 1])
def test_states_only_kernels_match_full_output(integ):
    """`return_forces=False` (training only consumes the states): identical states, no force tensors, gradients unchanged."""
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    B, T = 64, 120
    z = (syn.bump_terrain(syn.bump_params(9), 3.2, 0.1) * 0.3).to(DEV)
    mu = syn.wave_friction(3.2, 0.1).to(DEV)
    ctrl = syn.varying_controls(B, T, seed=3).to(DEV)
    grads = []
    outs = []
    for rf in (True, False):
        dp = make_dphysics(pts, masks, integ, 0.1, 3.2, return_forces=rf)
        zl = z.clone().requires_grad_(True)
        states, forces = dp(zl.unsqueeze(0), ctrl, friction=mu.unsqueeze(0))
        (states[0][:, ::7] ** 2).sum().backward()
        outs.append((states, forces)); grads.append(zl.grad)
    assert outs[1][1] == (None, None) and outs[0][1][0].shape == (B, T, 4, 3)
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b)
    assert hp.rel_err(grads[1].cpu(), grads[0].cpu()) <= 1e-5     # atomics order only
    # N = 33 (padded lane tile) and float64 / exact mode fall back to the full-output kernels transparently
    p33, m33 = syn.robot_points_box(33, seed=2, n_tracks=2)
    st, fo = make_dphysics(p33, m33, integ, 0.1, 3.2, return_forces=False)(z.unsqueeze(0), ctrl)
    assert fo == (None, None) and torch.isfinite(st[0]).all()
    st, fo = make_dphysics(pts, masks, integ, 0.1, 3.2, return_forces=False, precise=True)(z.unsqueeze(0), ctrl)
    assert fo[0] is not None


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('N', [4, 40])
def test_split_store_kernels_write_the_same_rows(integ, N):
    """Launches with a wave for every SIMD use the kernels whose state stores are split over the lanes of a group: same
    bits as the plain kernels (only the store instructions differ), with and without force outputs, with and without the
    unshifted-position buffer the backward needs."""
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_box(N, seed=N, n_tracks=2) if N > 4 else syn.robot_points_4()
    G = 4 if N == 4 else 64
    B_big = 1024 * 64 // G + 17                      # >= 1024 waves -> split-store kernels
    T = 24
    z = (syn.bump_terrain(syn.bump_params(5), 3.2, 0.1) * 0.3).to(DEV)
    ctrl = syn.varying_controls(B_big, T, seed=1).to(DEV)
    for forces in (True, False):
        for grad in (False, True):
            # (one point per lane in both launches: a small batch would otherwise take the component-parallel kernels)
            dp = make_dphysics(pts, masks, integ, 0.1, 3.2, return_forces=forces, points_per_lane=1 if N <= 4 else 0)
            zb = z.clone().requires_grad_(grad)
            big = dp(zb.unsqueeze(0), ctrl)
            zs = z.clone().requires_grad_(grad)
            small = dp(zs.unsqueeze(0), ctrl[:50].contiguous())
            for u, v in zip(big[0] + big[1], small[0] + small[1]):
                assert (u is None) == (v is None)
                if u is not None:
                    assert torch.equal(u[:50], v)
            if grad:            # the backward reads the rows (incl. the unshifted positions) the split kernels wrote
                big[0][0][:50].square().sum().backward()
                small[0][0].square().sum().backward()
                assert hp.rel_err(zb.grad.cpu(), zs.grad.cpu()) <= 1e-5


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('N,d_max', [(4, 3.2), (4, 0.6), (23, 1.0), (64, 3.2)])
def test_interleaved_map_kernels_read_the_same_cells(integ, N, d_max):
    """With one map pair shared by all rollouts the library reads an interleaved (z, mu) copy -- two 16-byte loads per contact
    point instead of eight 4-byte ones.  Same cells, same bits: full outputs (few waves and the split-store regime), states
    only, cost rows (raw and projected), with and without a friction map -- including rollouts that leave a small map, where
    the reference's clamp of the FLAT cell index folds neighbours onto cell 0 / HW-1 (d_max 0.6 and 1.0: 12 x 12 / 20 x 20 cells)."""
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_box(N, seed=N, n_tracks=2) if N > 4 else syn.robot_points_4()
    G = 4
    while G < N:
        G *= 2
    T = 300 if d_max < 3 else 40
    z = (syn.bump_terrain(syn.bump_params(7), d_max, 0.1) * 0.3).to(DEV)
    mu = (0.5 + 0.5 * torch.rand(z.shape, generator=torch.Generator().manual_seed(3))).to(DEV)
    for B in (37, 600 * 64 // G, 1024 * 64 // G + 5):    # no interleaving (few waves) | interleaved | interleaved + split stores
        ctrl = syn.varying_controls(B, T, seed=2).to(DEV)
        ctrl[:, :, 0] = ctrl[:, :, 0].abs() + 0.5                     # keep driving: on the small maps most rollouts leave the grid
        for friction in (mu, None):
            for forces in (True, False):
                outs = []
                for inter in (True, False):
                    dp = make_dphysics(pts, masks, integ, 0.1, d_max, return_forces=forces)
                    dp.interleave_maps = inter
                    st, fo = dp(z.unsqueeze(0), ctrl, friction=None if friction is None else friction.unsqueeze(0))
                    outs.append([o for o in list(st) + list(fo) if o is not None])
                if d_max < 3:     # the test is about leaving the map: make sure contact points do
                    assert float(outs[0][0][..., :2].abs().max()) > d_max - 0.1
                for u, v in zip(*outs):
                    assert torch.equal(u, v)
            for project in (True, False):
                rows = []
                for inter in (True, False):
                    dp = make_dphysics(pts, masks, integ, 0.1, d_max)
                    dp.interleave_maps = inter
                    r = dp.rollout_costs(z.unsqueeze(0), ctrl, friction=None if friction is None else friction.unsqueeze(0), pose_stride=10, project=project)
                    rows.append(r)
                for k in ('cost_rows', 'Xs', 'Rs', 'force_cost'):
                    assert torch.equal(rows[0][k], rows[1][k]), k


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_default_state_computed_in_the_kernel(dtype):
    """state=None: the rollout kernel builds the reference's default start (dphysics.py:554-559) itself and writes it back --
    same outputs as passing that state explicitly, the buffers it fills are what the backward pass then reads."""
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    B, T = 33, 60
    z = (syn.bump_terrain(syn.bump_params(4), 3.2, 0.1) * 0.3).to(DEV).to(dtype)
    ctrl = syn.varying_controls(B, T, seed=4).to(DEV).to(dtype)
    for integ in (0, 1):
        dp = make_dphysics(pts, masks, integ, 0.1, 3.2)
        za = z.clone().requires_grad_(True)
        a = dp(za.unsqueeze(0), ctrl)
        state = (torch.zeros(B, 3, device=DEV, dtype=dtype), torch.nn.functional.pad(ctrl[:, 0, 0:1], (0, 2)),
                 torch.eye(3, device=DEV, dtype=dtype).repeat(B, 1, 1), torch.nn.functional.pad(ctrl[:, 0, 1:2], (2, 0)))
        zb = z.clone().requires_grad_(True)
        b = dp(zb.unsqueeze(0), ctrl, state=state)
        for u, v in zip(a[0] + a[1], b[0] + b[1]):
            assert torch.equal(u, v)
        a[0][0].square().sum().backward()
        b[0][0].square().sum().backward()
        assert hp.rel_err(za.grad.cpu(), zb.grad.cpu()) <= (1e-5 if dtype == torch.float32 else 1e-10)
        ra = dp.rollout_costs(z.float().unsqueeze(0), ctrl.float(), pose_stride=10)
        rb = dp.rollout_costs(z.float().unsqueeze(0), ctrl.float(), state=tuple(t.float() for t in state), pose_stride=10)
        for k in ('cost_rows', 'Xs', 'Rs', 'force_cost'):
            assert torch.equal(ra[k], rb[k]), k


def test_time_constant_controls_are_read_in_place():
    """One (v, w) per rollout expanded over the horizon (what generate_controls / the planning node sample) is handed to the
    kernels as the [B,1,2] tensor it is (controls_stride_t = 0): same outputs as with the materialised [B,T,2] copy -- full
    outputs, both integrators, float32 and float64, default and explicit start state, and the path-cost mode; with autograd on
    the library gets a contiguous copy and gradients match."""
    from monoforce_amd import synthetic as syn
    from monoforce_amd.planner import sample_controls
    pts, masks = syn.robot_points_4()
    B, T = 41, 80
    z = (syn.bump_terrain(syn.bump_params(2), 3.2, 0.1) * 0.3).to(DEV)
    for integ in (0, 1):
        dp = make_dphysics(pts, masks, integ, 0.1, 3.2)
        dp.dphys_cfg.traj_sim_time = T * dp.dphys_cfg.dt + 1e-6
        dp = make_dphysics(pts, masks, integ, 0.1, 3.2)
        vw = torch.rand(B, 1, 2, generator=torch.Generator().manual_seed(integ)).to(DEV) * torch.tensor([1.0, 2.0], device=DEV) - torch.tensor([0.0, 1.0], device=DEV)
        Tn = int(dp.dphys_cfg.traj_sim_time / dp.dphys_cfg.dt)
        view = vw.expand(-1, Tn, -1)
        assert view.stride(1) == 0
        for dtype in (torch.float32, torch.float64):
            a = dp(z.to(dtype).unsqueeze(0), view.to(dtype) if dtype == torch.float32 else vw.to(dtype).expand(-1, Tn, -1))
            b = dp(z.to(dtype).unsqueeze(0), view.to(dtype).contiguous())
            for u, v in zip(a[0] + a[1], b[0] + b[1]):
                assert torch.equal(u, v)
        state = (torch.zeros(B, 3, device=DEV), torch.zeros(B, 3, device=DEV), torch.eye(3, device=DEV).repeat(B, 1, 1), torch.zeros(B, 3, device=DEV))
        ra = dp.rollout_costs(z.unsqueeze(0), view, state=tuple(t.clone() for t in state), pose_stride=10)
        rb = dp.rollout_costs(z.unsqueeze(0), view.contiguous(), state=tuple(t.clone() for t in state), pose_stride=10)
        rc = dp.rollout_costs(z.unsqueeze(0), view, pose_stride=10)
        rd = dp.rollout_costs(z.unsqueeze(0), view.contiguous(), pose_stride=10)
        for k in ('cost_rows', 'Xs', 'Rs', 'force_cost'):
            assert torch.equal(ra[k], rb[k]) and torch.equal(rc[k], rd[k]), k
        za, zb = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
        dp(za.unsqueeze(0), view)[0][0].square().sum().backward()
        dp(zb.unsqueeze(0), view.contiguous())[0][0].square().sum().backward()
        assert hp.rel_err(za.grad.cpu(), zb.grad.cpu()) <= 1e-5
    c = sample_controls(64, dp.dphys_cfg, DEV)
    assert c.shape == (64, Tn, 2) and c.stride(1) == 0


def test_very_large_batches_go_out_in_several_launches():
    """More than 2048 waves of rollouts are launched in chunks (two waves per SIMD is where these kernels peak): every rollout
    computes what it computes alone, whatever chunk it lands in -- full outputs, states only and cost rows, across a chunk edge."""
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_box(40, seed=40, n_tracks=2)       # 64 lanes per rollout: 2100 rollouts = 2100 waves, 2 launches
    B, T = 2100, 10
    z = (syn.bump_terrain(syn.bump_params(6), 3.2, 0.1) * 0.3).to(DEV)
    ctrl = syn.varying_controls(B, T, seed=3).to(DEV)
    tail = ctrl[2030:2100].contiguous()                               # rollouts on both sides of the edge at 2048
    for integ in (0, 1):
        for forces in (True, False):
            dp = make_dphysics(pts, masks, integ, 0.1, 3.2, return_forces=forces)
            big = dp(z.unsqueeze(0), ctrl)
            small = dp(z.unsqueeze(0), tail)
            for u, v in zip(big[0] + big[1], small[0] + small[1]):
                assert (u is None) == (v is None)
                if u is not None:
                    assert torch.equal(u[2030:2100], v)
        dp = make_dphysics(pts, masks, integ, 0.1, 3.2)
        rb = dp.rollout_costs(z.unsqueeze(0), ctrl, pose_stride=5)
        rs = dp.rollout_costs(z.unsqueeze(0), tail, pose_stride=5)
        for k in ('cost_rows', 'Xs', 'Rs', 'force_cost'):
            assert torch.equal(rb[k][2030:2100], rs[k]), k
