"""The fused staging kernel (terrain = geom - diff, both poolings, the interleaved pair) vs the reference's separate steps
(lss.py:158 + torch.nn.AvgPool2d, scripts/train.py:93-99,233-235), forward and backward, and its hand-off to the rollout."""
import pytest
import torch

from tests import helpers as hp

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('B,H,W,k', [(1, 256, 256, 1), (2, 128, 128, 4), (3, 50, 37, 4), (1, 64, 64, 8), (2, 33, 65, 2)])
def test_stage_terrain_matches_separate_ops(B, H, W, k):
    from monoforce_amd.terrain_stage import stage_terrain, staged_pair
    g = torch.Generator().manual_seed(B * 1000 + H + k)
    mk = lambda: torch.randn(B, 1, H, W, generator=g).to(DEV).requires_grad_(True)  # noqa: E731
    geom, diff, fric = mk(), mk(), mk()
    terrain, z, mu = stage_terrain(geom, diff, fric, k)
    pool = torch.nn.AvgPool2d(k, k) if k > 1 else torch.nn.Identity()
    geom2, diff2, fric2 = (t.detach().clone().requires_grad_(True) for t in (geom, diff, fric))
    t_ref = geom2 - diff2
    z_ref, mu_ref = pool(t_ref).squeeze(1), pool(fric2).squeeze(1)
    assert terrain.shape == t_ref.shape and z.shape == z_ref.shape and mu.shape == mu_ref.shape
    assert torch.equal(terrain, t_ref)
    assert hp.rel_err(z, z_ref) <= 1e-6 and hp.rel_err(mu, mu_ref) <= 1e-6
    zmu = staged_pair(z, mu)
    assert zmu is not None and torch.equal(zmu[..., 0], z) and torch.equal(zmu[..., 1], mu)
    assert staged_pair(z, mu.clone()) is None                       # another friction tensor: not the staged pair
    wt, wz, wm = torch.randn_like(terrain), torch.randn_like(z), torch.randn_like(mu)
    ((terrain * wt).sum() + (z * wz).sum() + (mu * wm).sum()).backward()
    ((t_ref * wt).sum() + (z_ref * wz).sum() + (mu_ref * wm).sum()).backward()
    for a, b in ((geom, geom2), (diff, diff2), (fric, fric2)):
        assert hp.rel_err(a.grad, b.grad) <= 1e-6
    # only one of the three outputs used: the others arrive as None in the backward
    geom3, diff3, fric3 = (t.detach().clone().requires_grad_(True) for t in (geom, diff, fric))
    _, z3, _ = stage_terrain(geom3, diff3, fric3, k)
    (z3 * wz).sum().backward()
    geom4, diff4 = (t.detach().clone().requires_grad_(True) for t in (geom, diff))
    (pool(geom4 - diff4).squeeze(1) * wz).sum().backward()
    assert hp.rel_err(geom3.grad, geom4.grad) <= 1e-6 and float(fric3.grad.abs().max()) == 0.0


@pytest.mark.parametrize('B', [64, 1024, 4096])
def test_rollout_reads_the_staged_pair(B):
    """A rollout fed the staged (z, mu) gives the same bits as one fed equal plain tensors, at batch sizes on both sides of the
    lane-mapping switches, forward and gradients to the head outputs."""
    from monoforce_amd import synthetic as syn
    from monoforce_amd.terrain_stage import stage_terrain
    from tests.test_rollout_gpu import make_dphysics
    pts, masks = syn.robot_points_4()
    T = 60
    geom = (syn.bump_terrain(syn.bump_params(3), 6.4, 0.05) + 0.3).to(DEV).view(1, 1, 256, 256).requires_grad_(True)
    diff = torch.full((1, 1, 256, 256), 0.3, device=DEV).requires_grad_(True)
    fric = syn.wave_friction(6.4, 0.05).to(DEV).view(1, 1, 256, 256).requires_grad_(True)
    ctrl = syn.const_controls(B, T, seed=4).to(DEV)
    dp = make_dphysics(pts, masks, 1, 0.05, 6.4)
    terrain, z, mu = stage_terrain(geom, diff, fric, 1)
    (Xs, _, Rs, _), (Fs, _) = dp(z, ctrl, friction=mu)
    assert dp.staged_handoffs == 1              # the rollout did take the interleaved pair
    (Xs[:, ::5].square().sum()).backward()
    geom2 = geom.detach().clone().requires_grad_(True)
    fric2 = fric.detach().clone().requires_grad_(True)
    z2, mu2 = (geom2 - diff.detach()).squeeze(1), fric2.squeeze(1)
    (Xs2, _, Rs2, _), (Fs2, _) = dp(z2, ctrl, friction=mu2)
    (Xs2[:, ::5].square().sum()).backward()
    assert dp.staged_handoffs == 1              # ... and the plain tensors did not
    assert torch.equal(Xs, Xs2) and torch.equal(Rs, Rs2) and torch.equal(Fs, Fs2)
    assert hp.rel_err(geom.grad, geom2.grad) <= 1e-5 and hp.rel_err(fric.grad, fric2.grad) <= 1e-5
