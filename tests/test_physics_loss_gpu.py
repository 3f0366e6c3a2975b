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
    from monoforce_amd.losses import physics_loss_aten as physics_loss, physics_loss_fused
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
    from monoforce_amd.losses import physics_loss_aten as physics_loss, physics_loss_fused
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


# ---- physics_loss inside the rollout's own launches (MfRolloutLoss; DPhysics.physics_loss_rollout) ----------------
def _fit_problem(B, T, loss_in_kernel, graph=False, res=0.1, d_max=3.2, gt_every=10, in_forward=False, value='backward', integ=1):
    from monoforce_amd import synthetic as syn
    from monoforce_amd.train import TerrainFitProblem
    from tests.test_rollout_gpu import make_dphysics
    pts, masks = syn.robot_points_4()
    dp = make_dphysics(pts, masks, integ, res, d_max)
    dp.loss_in_forward = in_forward      # True: the forward rollout kernel accumulates the loss itself; False: one small launch on its rows
    z_true = (syn.bump_terrain(syn.bump_params(3), d_max, res) * 0.3).to(DEV)
    mu = syn.wave_friction(d_max, res).to(DEV)
    ctrl = syn.const_controls(B, T, seed=2).to(DEV)
    prob = TerrainFitProblem(dp, z_true, mu, ctrl, gt_every=gt_every, graph=graph, loss_in_kernel=loss_in_kernel)
    # where the loss VALUE is formed: by the backward launch (the fit step's default), by the forward rollout kernel (in_forward), or
    # by one small launch on the forward's rows (value = 'launch')
    prob.loss_value_in_backward = value == 'backward' and not in_forward
    z = (z_true * 0.5).clone().requires_grad_(True)
    m = mu.clone().requires_grad_(True)
    return prob, z, m


@pytest.mark.parametrize('in_forward,value,integ', [(False, 'launch', 1), (True, 'forward', 1), (False, 'backward', 1),
                                                     (False, 'launch', 0), (False, 'backward', 0)])      # dynamics(): the backward half (round 4)
@pytest.mark.parametrize('B,T,gt_every', [(48, 300, 10), (37, 95, 10), (3, 41, 1), (1, 12, 5), (1500, 120, 10)])
def test_loss_inside_the_rollout_kernels_equals_the_two_kernel_route(B, T, gt_every, in_forward, value, integ):
    """Forward: the mean the rollout kernel finishes itself == mf_physics_loss_value_* on its outputs (and == the plain-torch
    restatement of losses.py:102-127); backward: dL/dXs formed by the fetching waves == the gradient rows mf_physics_loss_bwd_* writes,
    so the terrain / friction gradients agree to the rounding of the atomics' arrival order.  Batches with a trailing partial
    workgroup (37, 3, 1 rollouts), every row stamped (gt_every = 1), two workgroups per CU (1500).  Both forms of the forward half:
    the rollout kernel accumulating the loss itself (`loss_in_forward`, the LOSS kernels) and one small launch on the rows it wrote; the
    backward half is the same -- and, third form (the fit step's default), the backward launch forming the VALUE as well
    (MF_LOSS_VALUE_IN_BACKWARD)."""
    from monoforce_amd.losses import physics_loss_aten as physics_loss      # (the reference formulation in ATen ops: the referee)
    out = []
    for in_kernel in (False, True):
        prob, z, m = _fit_problem(B, T, in_kernel, gt_every=gt_every, in_forward=in_forward, value=value, integ=integ)
        vals = []
        for _ in range(3):                                   # launch after launch: the ticket comes back to zero
            loss = prob.step(z, m)
            vals.append(float(loss))
        assert len(set(vals)) == 1, vals
        out.append((vals[0], z.grad.clone(), m.grad.clone(), prob))
    assert out[1][3].loss_in_kernel and out[1][3].spec.fusable
    if integ == 0 and B > 1024:      # dynamics() streams up to 256 workgroups: beyond, the same value and gradient through the unfused route (up to 4096 rollouts)
        from monoforce_amd import _lib
        import ctypes as C
        d = _lib.MfRolloutDesc(B=B, T=T, N=4, H=64, W=64, integrator=0, math_mode=_lib.MF_MATH_FAST, force_stride=4, map_shared=1, layout=_lib.MF_LAYOUT_TIME_MAJOR)
        assert not _lib.lib().mf_rollout_loss_fusable(C.byref(d))
    assert abs(out[0][0] - out[1][0]) <= 2e-6 * abs(out[0][0]), (out[0][0], out[1][0])
    assert hp.rel_err(out[1][1].cpu(), out[0][1].cpu()) <= 2e-5 and hp.rel_err(out[1][2].cpu(), out[0][2].cpu()) <= 2e-5
    # ... and the value against the reference formula in plain torch on the rollout's outputs
    prob = out[1][3]
    with torch.no_grad():
        states, _ = prob.dp(z.detach().unsqueeze(0), prob.controls, friction=m.detach().unsqueeze(0))
        ref = physics_loss(states, prob.states_gt, prob.pred_ts, prob.gt_ts, nearest=prob.nearest.long())
    assert abs(float(ref) - out[1][0]) <= 2e-6 * abs(float(ref))


def test_loss_inside_the_kernels_replayed_as_a_graph_and_non_unit_upstream():
    """The four-launch step captured as ONE hipGraph replays to the same loss and gradients; a scaled loss scales the gradients
    (the upstream scalar reaches the fetching waves through MfRolloutLoss.gloss)."""
    prob, z, m = _fit_problem(256, 150, True)
    l0 = float(prob.step(z, m)); g0 = (z.grad.clone(), m.grad.clone())
    pg, zg, mg = _fit_problem(256, 150, True, graph=True, in_forward=True)
    for _ in range(3):
        lg = float(pg.step(zg, mg))
        assert pg.graph and abs(lg - l0) <= 1e-6 * abs(l0)
        assert hp.rel_err(zg.grad.cpu(), g0[0].cpu()) <= 1e-5 and hp.rel_err(mg.grad.cpu(), g0[1].cpu()) <= 1e-5
    z.grad = None; m.grad = None
    loss, _ = prob.dp.physics_loss_rollout(z.unsqueeze(0), prob.controls, prob.states_gt[0], prob.spec, friction=m.unsqueeze(0))
    (loss * 3.0).backward()
    assert hp.rel_err(z.grad.cpu(), (g0[0] * 3.0).cpu()) <= 1e-5


def test_loss_value_formed_by_the_backward_launch():
    """MF_LOSS_VALUE_IN_BACKWARD through `physics_loss_rollout(value_in_backward=True)`: the scalar is NaN between the forward and
    the backward (not a stale or uninitialised number), afterwards the value of the separate launch to rounding; the gradients do
    not depend on which launch forms the value; without `requires_grad` inputs (no backward to come) the forward's own route runs."""
    prob, z, m = _fit_problem(1024, 200, True, value='launch')
    args = (z.unsqueeze(0), prob.controls, prob.states_gt[0], prob.spec)
    res = {}
    for mode in (False, True):
        z.grad = None; m.grad = None
        loss, _ = prob.dp.physics_loss_rollout(*args, friction=m.unsqueeze(0), value_in_backward=mode)
        torch.cuda.synchronize()
        before = float(loss)
        assert (before != before) == mode, (mode, before)      # NaN exactly when the backward is to form it
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (float(loss), z.grad.clone(), m.grad.clone())
    assert abs(res[True][0] - res[False][0]) <= 2e-6 * abs(res[False][0]), (res[True][0], res[False][0])
    # (the pool's atomics arrive in a launch-dependent order: equal to rounding, not bit for bit)
    assert hp.rel_err(res[True][1].cpu(), res[False][1].cpu()) <= 1e-5 and hp.rel_err(res[True][2].cpu(), res[False][2].cpu()) <= 1e-5
    with torch.no_grad():
        loss, _ = prob.dp.physics_loss_rollout(z.detach().unsqueeze(0), prob.controls, prob.states_gt[0], prob.spec, friction=m.detach().unsqueeze(0),
                                               value_in_backward=True)
    assert abs(float(loss) - res[False][0]) <= 2e-6 * abs(res[False][0])


def test_loss_rollout_falls_back_where_the_library_cannot_fuse():
    """Exact arithmetic, float64 (outside the validation build), two stamps on one output row: `physics_loss_rollout` returns the same
    value and gradient through the unfused route (`mf_rollout_loss_fusable` / LossSpec.fusable say no); dynamics() (first case) fuses
    its backward half since round 4 and must agree all the same."""
    from monoforce_amd import synthetic as syn
    from monoforce_amd.losses import physics_loss_aten as physics_loss      # (the reference formulation in ATen ops: the referee)
    from tests.test_rollout_gpu import make_dphysics
    pts, masks = syn.robot_points_4()
    B, T = 16, 60
    z = (syn.bump_terrain(syn.bump_params(3), 3.2, 0.1) * 0.3)
    ctrl = syn.const_controls(B, T, seed=2)
    gen = torch.Generator().manual_seed(0)
    for kw, integ, dtype, crowded in (({}, 0, torch.float32, False), ({'precise': True}, 1, torch.float32, False),
                                      ({}, 1, torch.float64, False), ({}, 1, torch.float32, True)):
        dp = make_dphysics(pts, masks, integ, 0.1, 3.2, **kw)
        ts = torch.linspace(0, dp.dphys_cfg.traj_sim_time, int(dp.dphys_cfg.traj_sim_time / dp.dphys_cfg.dt))[:T]
        gt_ts = (ts[9::10] if not crowded else torch.tensor([0.091, 0.092, 0.3]))         # crowded: two stamps nearest to row 9
        Xgt = torch.randn(B, gt_ts.numel(), 3, generator=gen).to(DEV, dtype)
        spec = dp.loss_spec(gt_ts, gamma=0.9, n_steps=T)
        assert spec.fusable == (not crowded)
        zl = z.to(DEV, dtype).requires_grad_(True)
        loss, states = dp.physics_loss_rollout(zl.unsqueeze(0), ctrl.to(DEV, dtype), Xgt, spec)
        loss.backward()
        z2 = z.to(DEV, dtype).requires_grad_(True)
        st2, _ = dp(z2.unsqueeze(0), ctrl.to(DEV, dtype))
        ref = physics_loss(st2, [Xgt], ts.to(DEV, dtype).unsqueeze(0).expand(B, -1), gt_ts.to(DEV, dtype).unsqueeze(0).expand(B, -1))
        ref.backward()
        tol = 1e-5 if dtype == torch.float32 else 1e-10
        assert abs(float(loss) - float(ref)) <= tol * abs(float(ref)), (kw, integ, dtype, crowded)
        assert hp.rel_err(zl.grad.cpu(), z2.grad.cpu()) <= 10 * tol


@pytest.mark.parametrize('in_forward,value', [(False, 'launch'), (True, 'forward'), (False, 'backward')])
def test_non_finite_rows_behind_the_last_stamp_do_not_poison_the_fused_loss(in_forward, value):
    """the fused loss formed its term at EVERY row with weight 0 off the stamps, so a non-finite position on an unstamped
    row (a rollout that diverged after its last stamp) turned inf * 0 into a NaN gradient and value, where the unfused route and the
    reference (losses.py:116-127: only the stamped rows are gathered) never look at those rows.  Rows are masked by the stamp now.
    The kernels' clamps keep a rollout finite even on NaN terrain, so the rows are poisoned by hand: the backward launch re-reads
    the forward's Xs rows (documented as read-only for that reason), and everything behind the last stamp is set to inf before it."""
    from monoforce_amd import synthetic as syn
    from tests.test_rollout_gpu import make_dphysics
    pts, masks = syn.robot_points_4()
    B, T, res, d_max = 24, 300, 0.1, 3.2
    z = (syn.bump_terrain(syn.bump_params(3), d_max, res) * 0.3)
    ctrl = syn.const_controls(B, T, seed=2).to(DEV)
    dp = make_dphysics(pts, masks, 1, res, d_max, return_forces=False)
    dp.loss_in_forward = in_forward
    ts = torch.linspace(0, dp.dphys_cfg.traj_sim_time, int(dp.dphys_cfg.traj_sim_time / dp.dphys_cfg.dt))[:T]
    gt_ts = ts[9:60:10]                                                 # stamps within the first 0.6 s only: rows 9 .. 59
    Xgt = torch.randn(B, gt_ts.numel(), 3, generator=torch.Generator().manual_seed(0)).to(DEV) * 0.1
    spec = dp.loss_spec(gt_ts, gamma=0.9, n_steps=T)
    res_ = {}
    for poisoned in (False, True):
        zl = z.to(DEV).requires_grad_(True)
        loss, states = dp.physics_loss_rollout(zl.unsqueeze(0), ctrl, Xgt, spec, value_in_backward=(value == 'backward'))
        assert type(loss.grad_fn).__name__.startswith('_RolloutLossFn')
        if poisoned:
            states[0][:, 60:] = float('inf')                            # [B, T, 3] view of the time-major rows the backward reads
            states[0][:, 1:9] = float('nan')                            # ... and a few unstamped rows in front of the first stamp
        loss.backward()
        torch.cuda.synchronize()
        res_[poisoned] = (float(loss), zl.grad.clone())
    assert np.isfinite(res_[True][0]) and abs(res_[True][0] - res_[False][0]) <= 1e-6 * abs(res_[False][0]), (res_[True][0], res_[False][0])
    assert torch.isfinite(res_[True][1]).all()
    assert hp.rel_err(res_[True][1].cpu(), res_[False][1].cpu()) <= 1e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('expanded', [True, False])
def test_public_physics_loss_on_gpu_tensors_is_the_hip_loss_and_equals_the_aten_form(dtype, expanded):
    """`monoforce.losses.physics_loss` (the reference's import path) on GPU tensors: value and gradient of the reference formulation
    (`physics_loss_aten`, losses.py:102-127), through `mf_nearest_steps_*` + `mf_physics_loss_*` -- including irregular stamps, ties
    between two predicted steps (the first minimum wins, like torch.argmin) and stamps outside the predicted range."""
    from monoforce.losses import physics_loss
    from monoforce_amd import losses as L
    B, T1, T2 = 37, 120, 23
    g = torch.Generator().manual_seed(5)
    pred_row = torch.cumsum(torch.rand(T1, generator=g, dtype=torch.float64) * 0.02 + 0.001, 0)
    pred = (pred_row.unsqueeze(0).expand(B, -1) if expanded else pred_row.unsqueeze(0) + torch.rand(B, 1, generator=g, dtype=torch.float64) * 0.01)
    gt = torch.sort(torch.rand(B, T2, generator=g, dtype=torch.float64) * float(pred_row[-1]) * 1.1 - 0.05, dim=1).values
    gt[:, 3] = (pred[:, 10] + pred[:, 11]) / 2                      # a tie (exact in float64; float32 rounds it to one side or the other)
    gt[:, 4] = pred[:, 40]                                          # exactly on a step
    pred, gt = pred.to(DEV, dtype), gt.to(DEV, dtype)
    X = torch.randn(B, T1, 3, generator=g, dtype=torch.float64).to(DEV, dtype)
    Xgt = torch.randn(B, T2, 3, generator=g, dtype=torch.float64).to(DEV, dtype)
    assert torch.equal(L.nearest_steps_hip(pred, gt).long(), L.nearest_steps(pred, gt))
    out = []
    for fn in (physics_loss, L.physics_loss_aten):
        Xl = X.clone().transpose(0, 1).contiguous().transpose(0, 1).requires_grad_(True)      # the rollout's time-major layout as a [B,T,3] view
        loss = fn([Xl], [Xgt], pred, gt, gamma=0.9)
        loss.backward()
        out.append((float(loss), Xl.grad.clone()))
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    assert abs(out[0][0] - out[1][0]) <= tol * abs(out[1][0])
    assert hp.rel_err(out[0][1].cpu(), out[1][1].cpu()) <= tol
    assert type(physics_loss([X.requires_grad_(True)], [Xgt], pred, gt).grad_fn).__name__.startswith('_FusedPhysicsLoss')
    keep, L._HIP_LOSS = L._HIP_LOSS, False
    try:
        assert not type(physics_loss([X], [Xgt], pred, gt).grad_fn).__name__.startswith('_FusedPhysicsLoss')
    finally:
        L._HIP_LOSS = keep
