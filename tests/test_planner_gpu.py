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
    dp = type(dp)(dp.dphys_cfg, device=DEV)               # rebuild the time grid for the shorter horizon
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
    # flat ground: inclination cost ~ 0 for everyone
    if cost == 'inclination':
        flat = sh.shoot(torch.zeros_like(z), controls=c)
        assert float(flat['costs'].max()) < 5e-3 and float(flat['costs'].max()) < 0.2 * float(out['costs'].max())
