"""Fused physics loss (mf_physics_loss_*) vs the reference's golden vector and the plain-torch restatement."""
import numpy as np
import pytest
import torch

from tests import helpers as hp

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_fused_loss_matches_reference_golden():
    from monoforce_amd.losses import physics_loss_fused
    g = hp.load('physics_loss')
    X = torch.as_tensor(g['X']).to(DEV).requires_grad_(True)
    loss = physics_loss_fused([X], [torch.as_tensor(g['Xgt']).to(DEV)], torch.as_tensor(g['pred_ts']).to(DEV),
                              torch.as_tensor(g['gt_ts']).to(DEV), gamma=0.9)
    loss.backward()
    assert abs(float(loss) - float(g['loss'])) <= 1e-6 * abs(float(g['loss']))
    assert hp.rel_err(X.grad.cpu(), g['g_X']) <= 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_fused_loss_on_time_major_views_with_duplicate_stamps(dtype):
    """The rollout's outputs are [B,T,3] views of time-major buffers; several ground-truth stamps may share the nearest step."""
    from monoforce_amd.losses import physics_loss, physics_loss_fused
    B, T1, T2 = 37, 120, 11
    gen = torch.Generator().manual_seed(0)
    base = torch.randn(T1, B, 3, generator=gen, dtype=dtype).to(DEV)
    Xgt = torch.randn(B, T2, 3, generator=gen, dtype=dtype).to(DEV)
    pred_ts = torch.linspace(0, 5, 500, dtype=dtype)[:T1].to(DEV).unsqueeze(0).expand(B, -1)
    gt_ts = (torch.rand(B, T2, generator=gen, dtype=dtype) * 0.05).to(DEV)        # crowded: many duplicates of `nearest`
    res = []
    for fn in (physics_loss, physics_loss_fused):
        b = base.clone().requires_grad_(True)
        X = b.transpose(0, 1)
        loss = fn([X], [Xgt], pred_ts, gt_ts, gamma=0.9) * 3.0       # non-unit upstream gradient
        loss.backward()
        res.append((float(loss), b.grad.clone()))
    tol = 1e-6 if dtype == torch.float32 else 1e-13
    assert abs(res[0][0] - res[1][0]) <= tol * abs(res[0][0])
    assert hp.rel_err(res[1][1].cpu(), res[0][1].cpu()) <= tol * 10


def test_terrain_fit_step_same_gradients_with_fused_loss():
    from monoforce_amd import synthetic as syn
    from monoforce_amd.train import TerrainFitProblem
    from tests.test_rollout_gpu import make_dphysics
    pts, masks = syn.robot_points_4()
    dp = make_dphysics(pts, masks, 1, 0.1, 3.2)
    z_true = (syn.bump_terrain(syn.bump_params(3), 3.2, 0.1) * 0.3).to(DEV)
    mu = syn.wave_friction(3.2, 0.1).to(DEV)
    ctrl = syn.const_controls(48, 300, seed=2).to(DEV)
    out = []
    for fused in (False, True):
        prob = TerrainFitProblem(dp, z_true, mu, ctrl, fused_loss=fused)
        z = torch.zeros_like(z_true).requires_grad_(True); m = mu.clone().requires_grad_(True)
        loss = prob.step(z, m)
        out.append((float(loss), z.grad.clone(), m.grad.clone()))
    assert abs(out[0][0] - out[1][0]) <= 1e-5 * abs(out[0][0])
    assert hp.rel_err(out[1][1].cpu(), out[0][1].cpu()) <= 1e-4 and hp.rel_err(out[1][2].cpu(), out[0][2].cpu()) <= 1e-4


@pytest.mark.parametrize('B,T2', [(1, 1), (5, 50), (1024, 50), (3000, 7)])
def test_loss_value_is_finished_inside_the_launch_and_reusable(B, T2):
    """`mf_physics_loss_value_*`: the block taking the last ticket turns the per-block partial sums into the mean and resets the
    ticket -- the same value launch after launch (1 .. 587 blocks), equal to the plain-torch restatement of losses.py:102-127."""
    from monoforce_amd.losses import physics_loss, physics_loss_fused
    T1 = 10 * T2
    gen = torch.Generator().manual_seed(B)
    X = torch.randn(B, T1, 3, generator=gen).to(DEV)
    Xgt = torch.randn(B, T2, 3, generator=gen).to(DEV)
    pred_ts = (torch.arange(T1, dtype=torch.float32) * 0.01).unsqueeze(0).expand(B, -1).to(DEV)
    gt_ts = pred_ts[:, 9::10].contiguous()
    ref = physics_loss([X], [Xgt], pred_ts, gt_ts)
    vals = [float(physics_loss_fused([X], [Xgt], pred_ts, gt_ts)) for _ in range(4)]
    assert len(set(vals)) == 1, vals                      # deterministic, and the ticket came back to zero every time
    assert abs(vals[0] - float(ref)) <= 2e-6 * abs(float(ref))


def test_gradient_copy_pool_is_reused_clean():
    """The shared-map backward keeps its private gradient copies zeroed across steps (`mf_reduce_grad_copies_*` sums and clears in
    one launch): backward passes in a row through one module give the same gradients (to the rounding of the atomics' arrival
    order) -- also after a pass whose reduction never ran (the pool is found busy and refilled)."""
    from bench import build_problem
    from monoforce_amd import dphysics_bwd
    cfg, dp, pts, masks, z, mu, ctrl = build_problem(256, 120, 4, torch.device(DEV), 1, seed=0)
    zl, ml = z.to(DEV).clone().requires_grad_(True), mu.to(DEV).clone().requires_grad_(True)
    cd = ctrl.to(DEV)
    grads = []
    for it in range(4):
        zl.grad = None; ml.grad = None
        (Xs, _, _, _), _ = dp(zl.unsqueeze(0), cd, friction=ml.unsqueeze(0))
        Xs[:, 9::10].square().mean().backward()
        grads.append((zl.grad.clone(), ml.grad.clone()))
        if it == 1:                                        # leave dirt behind, as an exception between kernel and reduction would
            for p in dp._grad_pools.values():
                p.buf[:-16].fill_(7.0); p.busy = True
    assert float(grads[0][0].abs().max()) > 0
    for gz, gm in grads[1:]:
        assert hp.rel_err(gz, grads[0][0]) <= 1e-5 and hp.rel_err(gm, grads[0][1]) <= 1e-5
    pool = next(iter(dp._grad_pools.values()))
    assert float(pool.buf.abs().max()) == 0.0 and not pool.busy
