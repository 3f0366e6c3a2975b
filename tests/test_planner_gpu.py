"""Trajectory shooting (planner.py): sampled controls, path costs vs CPU re-computation, best-path selection."""
import numpy as np
import pytest
import torch

from tests import helpers as hp

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_inclination_cost_matches_scipy_euler():
    from scipy.spatial.transform import Rotation
    from monoforce_amd.planner import inclination_path_cost
    rng = np.random.RandomState(0)
    rpy = rng.uniform(-0.6, 0.6, (5, 40, 3))
    Rs = Rotation.from_euler('xyz', rpy.reshape(-1, 3)).as_matrix().reshape(5, 40, 3, 3)
    e = Rotation.from_matrix(Rs.reshape(-1, 3, 3)).as_euler('xyz').reshape(5, 40, 3)
    ref = np.abs(e[..., 0]).mean(-1) + np.abs(e[..., 1]).mean(-1)
    got = inclination_path_cost(torch.as_tensor(Rs)).numpy()
    assert np.allclose(got, ref, atol=1e-12)


@pytest.mark.parametrize('cost', ['inclination', 'force'])
def test_shooter_selects_cheapest_path(cost):
    from monoforce_amd import synthetic as syn
    from monoforce_amd.planner import TrajectoryShooter, force_path_cost, inclination_path_cost, sample_controls
    from tests.test_rollout_gpu import make_dphysics
    pts, masks = syn.robot_points_4()
    dp = make_dphysics(pts, masks, 1, 0.1, 6.4)
    dp.dphys_cfg.traj_sim_time = 2.0
    # one point per lane throughout: the path-cost kernels use that mapping, and poses are compared bit for bit below
    dp = type(dp)(dp.dphys_cfg, device=DEV, points_per_lane=1)               # rebuild the time grid for the shorter horizon
    z = (syn.bump_terrain(syn.bump_params(2), 6.4, 0.1) * 0.8).to(DEV)
    sh = TrajectoryShooter(dp, n_trajs=512, cost=cost)
    out = sh.shoot(z, generator=torch.Generator(device=DEV).manual_seed(1))
    c = out['controls']
    assert c.shape == (512, 200, 2) and bool((c[:256, 0, 0] >= 0.5).all()) and bool((c[256:, 0, 0] <= -0.5).all())
    assert bool((c[:, 0, 1].abs() <= 2.0).all()) and bool((c[:, 0] == c[:, -1]).all())
    # costs recomputed from a full-output rollout of the same controls
    (Xs, _, Rs, _), (Fs, _) = dp(z.unsqueeze(0), c)
    ref = (force_path_cost(Fs) if cost == 'force' else inclination_path_cost(Rs)).cpu()
    assert hp.rel_err(out['costs'].cpu(), ref) <= 1e-5
    assert out['best'] == int(torch.argmin(out['costs'])) and torch.isfinite(out['costs']).all()
    # the fused path keeps poses[::pose_stride] (0.5 s like the node) plus the final pose, bit-identical to the full rollout's
    steps = out['pose_steps'].cpu()
    assert steps.tolist() == [0, 50, 100, 150, 199] and out['Xs'].shape == (512, 5, 3) and out['Rs'].shape == (512, 5, 3, 3)
    assert torch.equal(out['Xs'].cpu(), Xs[:, steps].cpu()) and torch.equal(out['Rs'].cpu(), Rs[:, steps].cpu())
    # ... and the un-fused shooter (full output rows) gives the same costs
    out2 = TrajectoryShooter(dp, n_trajs=512, cost=cost, fused=False).shoot(z, controls=c)
    assert hp.rel_err(out2['costs'].cpu(), ref) <= 1e-6 and out2['Xs'].shape == (512, 200, 3)
    # flat ground: inclination cost ~ 0 for everyone
    if cost == 'inclination':
        flat = sh.shoot(torch.zeros_like(z), controls=c)
        assert float(flat['costs'].max()) < 5e-3 and float(flat['costs'].max()) < 0.2 * float(out['costs'].max())


def _start_state(B, seed=2):
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(seed)
    f = lambda a: torch.as_tensor(a, dtype=torch.float32)  # noqa: E731
    R = Rotation.from_euler('xyz', rng.uniform(-0.1, 0.1, (B, 3)) * [1, 1, 10]).as_matrix()
    return f(rng.uniform(-0.5, 0.5, (B, 3)) * [1, 1, 0.1]), f(rng.uniform(-0.3, 0.3, (B, 3))), f(R), f(rng.uniform(-0.2, 0.2, (B, 3)))


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('N,n_tracks,stride', [(4, 2, 1), (32, 4, 7), (175, 2, 50)])
def test_cost_rows_match_full_outputs(integ, N, n_tracks, stride):
    """Path-cost kernel vs the full-output kernel on the same inputs: cost rows, decimated poses, per-rollout maps, a start
    state, all lane mappings (one point per lane ... 64 x 4)."""
    from monoforce_amd import synthetic as syn
    from tests.test_rollout_gpu import make_dphysics
    pts, masks = syn.robot_points_box(N, seed=N, n_tracks=n_tracks) if N > 4 else syn.robot_points_4()
    B, T = 37, 120
    dp = make_dphysics(pts, masks, integ, 0.1, 3.2, points_per_lane=1 if N <= 4 else 0)   # the cost kernels' own mapping (bit equality below)
    z = torch.stack([syn.bump_terrain(syn.bump_params(3 + k % 3), 3.2, 0.1) * 0.5 for k in range(B)]).to(DEV)
    mu = torch.stack([syn.wave_friction(3.2, 0.1, 0.5, 1.0, 1.0 + 0.1 * k, 0.9) for k in range(B)]).to(DEV)
    ctrl = syn.varying_controls(B, T, seed=5).to(DEV)
    st = tuple(t.to(DEV) for t in _start_state(B))
    (Xs, _, Rs, _), (Fs, _) = dp(z, ctrl, state=tuple(t.clone() for t in st), friction=mu)
    out = dp.rollout_costs(z, ctrl, state=st, friction=mu, pose_stride=stride)
    rows = out['cost_rows']
    assert rows.shape == (B, T, 4)
    # Bit equality with the full-output kernel wherever both run the same step code.  A body over several waves (N > 64) under the
    # default integrator does not: its full-output forward is software-pipelined around ONE workgroup exchange per step (contact
    # count of step n + 1 with the wrench of step n, rollout_fwd_kernel.h), the cost kernel keeps the plain two-exchange step --
    # same formulas, FMA contraction chosen per code shape: equal to float32 rounding over the horizon instead
    same = torch.equal if not (N > 64 and integ == 1) else (lambda u, v: hp.rel_err(u, v) <= 2e-5)
    if integ == 0:          # dynamics(): R stays orthonormal and its third row is stored as is
        assert torch.equal(rows[..., :3], Rs[:, :, 2, :])
    else:                   # odeint-euler: third row of the nearest rotation (two Newton steps in the kernel vs SVD here)
        U, _, Vh = torch.linalg.svd(Rs.double().cpu())
        assert float((rows[..., :3].double().cpu() - (U @ Vh)[:, :, 2, :]).abs().max()) <= 2e-6
        raw = dp.rollout_costs(z, ctrl, state=st, friction=mu, pose_stride=stride, project=False)      # ... or the raw row
        assert same(raw['cost_rows'][..., :3], Rs[:, :, 2, :]) and torch.equal(raw['force_cost'], out['force_cost'])
        from monoforce_amd.planner import nearest_rotation_row2
        assert float((nearest_rotation_row2(Rs).double().cpu() - (U @ Vh)[:, :, 2, :]).abs().max()) <= 2e-6
    s_ref = torch.norm(Fs, dim=-1).std(dim=-1)
    ftol = 2e-6 if same is torch.equal else 1e-4
    assert hp.rel_err(rows[..., 3].cpu(), s_ref.cpu()) <= ftol, hp.rel_err(rows[..., 3].cpu(), s_ref.cpu())
    assert hp.rel_err(out['force_cost'].cpu(), s_ref.std(dim=-1).cpu()) <= 10 * ftol      # std over time, Welford in the kernel
    steps = out['pose_steps']
    assert steps[-1] == T - 1 and steps.numel() == 1 + -(-(T - 1) // stride)
    assert same(out['Xs'], Xs[:, steps]) and same(out['Rs'], Rs[:, steps])
    assert torch.equal(st[0].cpu(), _start_state(B)[0])      # the caller's start state is untouched (the snap works on a copy)


@pytest.mark.parametrize('name', ['A', 'B', 'C'])
@pytest.mark.parametrize('integ', [0, 1])
def test_path_costs_vs_reference_rollouts(name, integ):
    """The kernel's path costs against the reference nodes' formulas applied to the REFERENCE's own rollout outputs (golden
    vectors): norm(F_springs).std(points).std(time) (monoforce_node.py:91) and mean|roll| + mean|pitch| of scipy's
    `as_euler('xyz')` (diff_physics.py:263-266)."""
    from scipy.spatial.transform import Rotation
    from monoforce_amd.planner import costs_from_rows
    from tests.test_rollout_gpu import make_dphysics
    g = hp.load('rollout_small')
    pts, masks, z, ctrl, state, mu = hp.small_case(g, name, torch.float32)
    pre = f'{name}/f64/i{integ}/'
    Fs, Rs = g[pre + 'Fs'], g[pre + 'Rs']
    B, T = Fs.shape[:2]
    ref_force = np.linalg.norm(Fs, axis=-1).std(axis=-1, ddof=1).std(axis=-1, ddof=1)            # torch.std is unbiased
    rpy = Rotation.from_matrix(Rs.reshape(-1, 3, 3)).as_euler('xyz').reshape(B, T, 3)
    ref_incl = np.abs(rpy[..., 0]).mean(-1) + np.abs(rpy[..., 1]).mean(-1)
    dp = make_dphysics(pts, masks, integ, hp.SMALL['grid_res'], hp.SMALL['d_max'])
    st = None if state is None else tuple(s.to(DEV) for s in state)
    out = dp.rollout_costs(z.to(DEV), ctrl.to(DEV), state=st, friction=None if mu is None else mu.to(DEV), pose_stride=8)
    assert hp.rel_err(out['force_cost'].cpu().double(), ref_force) <= 1e-3, hp.rel_err(out['force_cost'].cpu().double(), ref_force)
    incl = costs_from_rows(out['cost_rows'], 'inclination').cpu().double()
    assert hp.rel_err(incl, ref_incl) <= 1e-3, hp.rel_err(incl, ref_incl)
    steps = out['pose_steps'].cpu().numpy()
    assert hp.rel_err(out['Xs'].cpu().double(), g[pre + 'Xs'][:, steps]) <= 1e-4
