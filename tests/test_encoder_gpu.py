"""Terrain encoder end to end on the GPU: the HIP splat inside LiftSplatShoot, and one full training step."""
import numpy as np
import pytest
import torch

from oracle import splat_oracle as so
from tests import helpers as hp

pytestmark = pytest.mark.gpu
DEV = 'cuda'
SMALL = dict(grid_conf=dict(xbound=[-3.2, 3.2, 0.1], ybound=[-3.2, 3.2, 0.1], zbound=[-2.0, 2.0, 4.0], dbound=[0.6, 3.4, 0.2]),
             data_aug_conf=dict(final_dim=(64, 96)))


def test_get_voxels_matches_oracle_on_lifted_features():
    from monoforce_amd import synthetic as syn
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    torch.manual_seed(0)
    m = LiftSplatShoot(SMALL['grid_conf'], SMALL['data_aug_conf']).to(DEV).eval()
    B = 2
    x = torch.randn(B, 3, 3, 64, 96, device=DEV)
    rig = [t.to(DEV) for t in syn.lss_camera_rig(B, 3, 64, 96, 40.0)]
    with torch.no_grad():
        geom = m.get_geometry(*rig)
        feats = m.get_cam_feats(x)
        # the plan of THIS geometry tensor (get_geometry ran torch.inverse + matmul on the device), so that get_voxels and the oracle
        # pool the same indices
        from monoforce_amd import splat
        bev = m.get_voxels(x, *rig, plan=splat.SplatPlan(geom, m.dx, m.bx, m.nx))
        bev_rig = m.get_voxels(x, *rig)       # the default route: the plan straight from the calibration tensors (in-kernel inverses)
        out = m(x, *rig)
    assert feats.shape == (B, 3, m.D, 4, 6, 64) and bev.shape == (B, 64, 64, 64)
    ref, kept = so.voxel_pooling(geom.cpu().numpy(), feats.cpu().numpy(), m.dx.cpu().numpy(), m.bx.cpu().numpy(), m.nx.cpu().numpy())
    assert kept.mean() > 0.3
    assert hp.rel_err(bev.cpu(), ref) <= 1e-6
    # this synthetic rig puts frustum points EXACTLY on voxel faces (focal 40, depths in multiples of 0.2, 0.1 m voxels: (76 - 48) / 40 * 2.0
    # = 1.4); the in-kernel 3 x 3 product rounds like torch's CPU matmul (unfused), the device matmul of get_geometry does not, so a
    # handful of such points land in the neighbouring voxel: a few voxel columns differ, everything else is the same sum
    cols = (bev_rig.cpu() - torch.as_tensor(ref)).abs().amax(dim=1) > 1e-6 * float(np.abs(ref).max())
    assert float(cols.float().mean()) <= 5e-3, float(cols.float().mean())
    assert set(out) == {'geom', 'terrain', 'diff', 'friction'} and out['terrain'].shape == (B, 1, 64, 64)
    assert all(torch.isfinite(v).all() for v in out.values())


def test_training_step_runs_and_learns():
    """Config-4 style step at reduced size: encoder -> shared predicted terrain -> 64 rollouts -> physics + height-map
    losses -> backward through the rollout and splat kernels -> Adam.  The loss must be finite and go down."""
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    from monoforce_amd import synthetic as syn
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    from monoforce_amd.train import EncoderTrainStep, synthetic_encoder_batch
    torch.manual_seed(0)
    gc = dict(xbound=[-3.2, 3.2, 0.1], ybound=[-3.2, 3.2, 0.1], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 3.4, 0.2])
    enc = LiftSplatShoot(gc, dict(final_dim=(64, 128))).to(DEV).train()
    pts, masks = syn.robot_points_4()
    cfg = DPhysConfig(robot='tradr', grid_res=0.1, robot_points=pts, driving_parts=masks)
    cfg.d_max, cfg.traj_sim_time = 3.2, 1.0
    dp = DPhysics(cfg, device=DEV)
    batch = synthetic_encoder_batch(enc, dp, n_rollouts=64, device=DEV, img_hw=(64, 128))
    step = EncoderTrainStep(enc, dp, lr=2e-4)
    losses = [float(step.step(batch)[0]) for _ in range(60)]
    assert all(np.isfinite(losses)), losses
    # Adam on one sample is noisy (single steps spike to 2-9x the running level at any time) and not bit-reproducible (MIOpen
    # algorithms, float atomics), but it fits: over 30 repeated runs the best loss of steps 30..39 was 0.02-0.28 of the first
    # (median 0.05), so the bar is the best loss after step 20 of 60 below half the first
    assert min(losses[20:]) < 0.5 * losses[0], losses
    g = [p.grad for p in enc.parameters() if p.requires_grad and p.grad is not None]
    assert len(g) > 100 and all(torch.isfinite(t).all() for t in g)
    assert any(float(p.grad.abs().max()) > 0 for n, p in enc.named_parameters() if n.startswith('camencode.depthnet'))


def test_compute_losses_on_the_references_sample_tuple():
    """the ROUGH sample tuple in the reference's order (datasets/rough.py:651-663) through
    `EncoderTrainStep.compute_losses` (scripts/train.py:377-410): same three losses as the package's own batch layout, and the
    reference's own case -- ONE trajectory per sample, per-sample predicted maps -- runs through the same entry."""
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    from monoforce_amd import synthetic as syn
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    from monoforce_amd.train import EncoderTrainStep, synthetic_encoder_batch, synthetic_rough_batch
    torch.manual_seed(0)
    gc = dict(xbound=[-3.2, 3.2, 0.1], ybound=[-3.2, 3.2, 0.1], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 3.4, 0.2])
    enc = LiftSplatShoot(gc, dict(final_dim=(64, 128))).to(DEV).eval()
    pts, masks = syn.robot_points_4()
    cfg = DPhysConfig(robot='tradr', grid_res=0.1, robot_points=pts, driving_parts=masks)
    cfg.d_max, cfg.traj_sim_time = 3.2, 1.0
    dp = DPhysics(cfg, device=DEV)
    step = EncoderTrainStep(enc, dp, lr=2e-4)
    b9 = synthetic_encoder_batch(enc, dp, n_rollouts=32, device=DEV, img_hw=(64, 128))
    b16 = synthetic_rough_batch(enc, dp, n_rollouts=32, device=DEV, img_hw=(64, 128))
    assert len(b16) == 16
    (imgs, rots, trans, intrins, post_rots, post_trans, hm_geom, hm_terrain, control_ts, controls, pose0, traj_ts, Xs, Xds, Rs, Om) = b16
    assert imgs.shape == (1, 4, 3, 64, 128) and hm_geom.shape == (1, 2, 64, 64) and controls.shape == (32, 100, 2) and pose0.shape == (32, 4, 4)
    assert control_ts.shape == (32, 100) and traj_ts.shape == (32, 10) and Xs.shape == (32, 10, 3) and Rs.shape == (32, 10, 3, 3)
    with torch.no_grad():
        a = [float(v) for v in step.losses(b9)]
        b = [float(v) for v in step.compute_losses(b16)]
    assert all(np.isfinite(a)) and all(abs(x - y) <= 1e-5 * max(abs(x), 1e-6) for x, y in zip(a, b)), (a, b)
    # the reference's collation: Bs samples, each with its own images, maps and ONE trajectory
    Bs = 3
    rep = lambda t: t.repeat(Bs, *([1] * (t.dim() - 1)))  # noqa: E731
    per_sample = (rep(imgs), rep(rots), rep(trans), rep(intrins), rep(post_rots), rep(post_trans), rep(hm_geom), rep(hm_terrain),
                  control_ts[:Bs], controls[:Bs], pose0[:Bs], traj_ts[:Bs], Xs[:Bs], Xds[:Bs], Rs[:Bs], Om[:Bs])
    with torch.no_grad():
        c = [float(v) for v in step.compute_losses(per_sample)]
    assert all(np.isfinite(c)) and abs(c[0] - a[0]) <= 1e-4 * abs(a[0])      # identical samples: the same height-map loss


def test_train_step_replayed_as_one_graph_equals_launch_by_launch():
    """`EncoderTrainStep(graph=True)`: encoder forward, lift-splat, heads, staging, rollout + physics loss, backward, gradient
    clipping and Adam captured once and replayed as ONE hipGraph launch.  In eval mode (no drop-connect randomness, batch norm on
    its running statistics) the replayed steps follow the launch-by-launch ones: same losses step by step, same parameters after."""
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    from monoforce_amd import synthetic as syn
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    from monoforce_amd.train import EncoderTrainStep, synthetic_encoder_batch
    gc = dict(xbound=[-3.2, 3.2, 0.1], ybound=[-3.2, 3.2, 0.1], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 3.4, 0.2])
    pts, masks = syn.robot_points_4()
    runs = []
    for graph in (False, True):
        torch.manual_seed(0)
        enc = LiftSplatShoot(gc, dict(final_dim=(64, 128))).to(DEV).eval()
        cfg = DPhysConfig(robot='tradr', grid_res=0.1, robot_points=pts, driving_parts=masks)
        cfg.d_max, cfg.traj_sim_time = 3.2, 1.0
        dp = DPhysics(cfg, device=DEV)
        batch = synthetic_encoder_batch(enc, dp, n_rollouts=64, device=DEV, img_hw=(64, 128))
        step = EncoderTrainStep(enc, dp, lr=2e-4, graph=graph)
        # graph mode: the first call's three warm-up steps leave no trace (parameters, batch-norm statistics and optimizer state are
        # restored before the capture), so replay k IS step k of the same optimisation
        losses = [float(step.step(batch)[0]) for _ in range(7)]
        assert not graph or (step.graph and step._cap is not None), 'the capture fell back to launch by launch'
        runs.append((losses, [p.detach().clone() for p in enc.parameters()]))
    eager, graphed = runs
    assert np.isfinite(eager[0]).all() and np.isfinite(graphed[0]).all()
    for k, lg in enumerate(graphed[0]):
        assert abs(lg - eager[0][k]) <= 2e-2 * abs(eager[0][k]), (k, lg, eager[0])
    assert abs(graphed[0][0] - eager[0][0]) <= 1e-4 * abs(eager[0][0])      # the first step starts from the SAME parameters
    # (Adam moves a parameter whose gradient is at the noise level of the float atomics by up to lr per step in EITHER direction:
    #  tensors that are still ~0 after 7 steps are held to that absolute bound, the others to 5 % of their largest entry)
    atol = 2 * 7 * 2e-4
    worst = max(float(((a - b).abs().max() - atol).clamp_min(0) / b.abs().max().clamp_min(1e-6)) for a, b in zip(graphed[1], eager[1]))
    assert worst <= 5e-2, worst


def _stamp_rig(graph=False, n_rollouts=32):
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    from monoforce_amd import synthetic as syn
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    from monoforce_amd.train import EncoderTrainStep, synthetic_rough_batch
    torch.manual_seed(0)
    gc = dict(xbound=[-3.2, 3.2, 0.1], ybound=[-3.2, 3.2, 0.1], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 3.4, 0.2])
    enc = LiftSplatShoot(gc, dict(final_dim=(64, 128))).to(DEV).eval()
    pts, masks = syn.robot_points_4()
    cfg = DPhysConfig(robot='tradr', grid_res=0.1, robot_points=pts, driving_parts=masks)
    cfg.d_max, cfg.traj_sim_time = 3.2, 1.0
    dp = DPhysics(cfg, device=DEV)
    return enc, dp, EncoderTrainStep(enc, dp, lr=2e-4, graph=graph), list(synthetic_rough_batch(enc, dp, n_rollouts=n_rollouts, device=DEV, img_hw=(64, 128)))


def _plain_physics_loss(dp, enc, b16):
    """The physics term through the module's plain entry points: `DPhysics.forward` + the torch `physics_loss` -- no fused tables."""
    from monoforce_amd.losses import physics_loss_aten as physics_loss      # (the reference formulation in ATen ops: the referee)
    (imgs, rots, trans, intrins, post_rots, post_trans, hm_geom, hm_terrain, control_ts, controls, pose0, traj_ts, Xs, Xds, Rs, Om) = b16
    with torch.no_grad():
        out = enc(imgs, rots, trans, intrins, post_rots, post_trans)
        k = max(int(round(dp.dphys_cfg.grid_res / float(enc.dx[0]))), 1)
        pool = torch.nn.AvgPool2d(k, k) if k > 1 else torch.nn.Identity()
        z, mu = pool(out['terrain']).squeeze(1), pool(out['friction']).squeeze(1)
        x0 = pose0[:, :3, 3].clone()
        st = (x0, torch.zeros_like(x0), pose0[:, :3, :3].contiguous(), torch.zeros_like(x0))
        states, _ = dp(z_grid=z, controls=controls, state=st, friction=mu)
        return float(physics_loss(states, [Xs, Xds, Rs, Om], control_ts, traj_ts))


@pytest.mark.parametrize('shared_rows', [True, False])
def test_stamp_tables_follow_the_batch_not_its_address(shared_rows):
    """the fused-loss tables and the nearest-step table used to be keyed on `data_ptr()`; a later batch at the
    same address (the caching allocator recycles it; a fixed-rig loop copies into the same tensors) was scored against the FIRST
    batch's stamps.  Three batches through one `EncoderTrainStep`: the original, new stamps copied into the same tensors, and
    new stamps in new tensors -- each must equal the plain `physics_loss` of ITS stamps."""
    enc, dp, step, b = _stamp_rig()
    TS, X = 11, 12      # traj_ts, Xs in the reference's 16-tuple
    if not shared_rows:                     # per-rollout stamps: the unfused route and its nearest-step table
        b[TS] = b[TS] + 0.002 * torch.arange(b[TS].shape[0], device=DEV).unsqueeze(1)
    with torch.no_grad():
        l0 = float(step.compute_losses(tuple(b))[2])
    assert (step._loss_spec(b[TS], b[9].shape[1]) is not None) == shared_rows
    assert abs(l0 - _plain_physics_loss(dp, enc, b)) <= 1e-5 * abs(l0)
    ptr = b[TS].data_ptr()
    b[TS].mul_(0.5).add_(0.013)             # the same tensors, other stamp times (other rows, other weights)
    b[X].add_(0.05)
    with torch.no_grad():
        l1 = float(step.compute_losses(tuple(b))[2])
    assert b[TS].data_ptr() == ptr and abs(l1 - l0) > 1e-3 * abs(l0)
    assert abs(l1 - _plain_physics_loss(dp, enc, b)) <= 1e-5 * abs(l1)
    b[TS] = (b[TS] * 1.7 + 0.004).contiguous()      # new tensors (possibly at a recycled address)
    b[X] = b[X] - 0.03
    with torch.no_grad():
        l2 = float(step.compute_losses(tuple(b))[2])
    assert abs(l2 - _plain_physics_loss(dp, enc, b)) <= 1e-5 * abs(l2)
    assert len(step._specs) <= 8


def test_stamps_the_fused_loss_cannot_carry_fail_loudly():
    """A later batch of a structure that took the fused route whose stamps put two of them on ONE output row: NaN, not a loss
    against stale tables."""
    enc, dp, step, b = _stamp_rig()
    with torch.no_grad():
        assert np.isfinite(float(step.compute_losses(tuple(b))[2]))
        b[11][:, 1] = b[11][:, 0] + 1e-4            # stamps 0 and 1 now share their nearest output row
        assert np.isnan(float(step.compute_losses(tuple(b))[2]))


def test_replayed_step_reads_the_stamps_copied_into_its_batch():
    """ADVICE r3 (medium, second half): the captured step rebuilds its stamp tables on the device, so a fixed-rig loop that copies
    the next sample (stamps included) into the batch tensors gets that sample's loss from the replay."""
    enc, dp, step, b = _stamp_rig(graph=True)
    from monoforce_amd.train import synthetic_encoder_batch          # (the 9-tuple layout `step()` takes)
    b9 = synthetic_encoder_batch(enc, dp, n_rollouts=32, device=DEV, img_hw=(64, 128))
    (inputs, hm_geom, hm_terrain, controls, pose0, states_gt, pred_ts, gt_ts, nearest) = b9
    step.opt.param_groups[0]['lr'] = 0.0            # the parameters stay put: every replay scores the same encoder
    step.w = (0.0, 0.0, 1.0)
    l0 = float(step.step(b9)[0])
    assert step.graph and step._cap is not None
    gt_ts.mul_(0.5).add_(0.013)
    states_gt[0].add_(0.05)
    l1 = float(step.step(b9)[0])                    # a replay
    ref = EncoderTrainStepRef(enc, dp, b9)
    assert abs(l1 - l0) > 1e-3 * abs(l0) and abs(l1 - ref) <= 1e-4 * abs(ref), (l0, l1, ref)


def EncoderTrainStepRef(enc, dp, b9):
    from monoforce_amd.train import EncoderTrainStep
    fresh = EncoderTrainStep(enc, dp, lr=0.0)
    fresh.loss_in_kernel = False                    # plain route: DPhysics.forward + mf_physics_loss_* on freshly computed nearest steps
    from monoforce_amd.losses import nearest_steps
    (inputs, hm_geom, hm_terrain, controls, pose0, states_gt, pred_ts, gt_ts, _) = b9
    with torch.no_grad():
        return float(fresh.losses((inputs, hm_geom, hm_terrain, controls, pose0, states_gt, pred_ts, gt_ts, nearest_steps(pred_ts, gt_ts).to(torch.int32)))[2])


@pytest.mark.parametrize('k,stride', [(3, 1), (5, 1), (3, 2), (5, 2)])
def test_depthwise_convolution_off_miopen_matches_the_library_route(k, stride, monkeypatch):
    """backbones.Conv2dStaticSame sends depthwise convolutions to ATen's depthwise kernels (MIOpen's immediate mode answers
    these shapes with naive_conv): values and both gradients against the MIOpen route of the same module."""
    from monoforce_amd import backbones as bb
    torch.manual_seed(k * 10 + stride)
    conv = bb.Conv2dStaticSame(48, 48, k, stride=stride, groups=48).cuda()
    x = torch.randn(4, 48, 32, 64, device='cuda')
    out = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('MF_BACKBONE_LEAN', flag)
        xi = x.clone().requires_grad_(True)
        conv.zero_grad()
        y = conv(xi)
        (y * torch.linspace(-1, 1, y.numel(), device='cuda').view_as(y)).sum().backward()
        out[flag] = (y.detach(), xi.grad.clone(), conv.weight.grad.clone())
    for a, b in zip(out['1'], out['0']):
        assert a.shape == b.shape
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max()))


def test_two_forwards_before_one_backward_without_the_plan_cache():
    """`cache_plan = False` rebuilds the splat plan every forward in a persistent workspace: a SECOND forward with another
    camera rig before the first forward's backward must not overwrite the plan that backward will scatter with.  Two views summed into one
    loss: gradients equal the cached-plan run's; the two plans live in different workspaces; the slot is reused once its graph is gone."""
    from monoforce_amd import synthetic as syn
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    torch.manual_seed(0)
    m = LiftSplatShoot(SMALL['grid_conf'], SMALL['data_aug_conf']).to(DEV).eval()
    x1, x2 = torch.randn(1, 3, 3, 64, 96, device=DEV), torch.randn(1, 3, 3, 64, 96, device=DEV)
    rig1 = [t.to(DEV) for t in syn.lss_camera_rig(1, 3, 64, 96, 40.0)]
    rig2 = [t.clone() for t in rig1]
    rig2[1] = rig2[1] + torch.tensor([0.35, -0.2, 0.0], device=DEV)          # the second view: the rig moved -> other voxels
    w1, w2 = torch.randn(1, 64, 64, 64, device=DEV), torch.randn(1, 64, 64, 64, device=DEV)
    params = [p for p in m.camencode.parameters() if p.requires_grad]

    def grads(cache):
        m.cache_plan = cache
        m._plan_cache = None
        for p in params:
            p.grad = None
        b1 = m.get_voxels(x1, *rig1)
        b2 = m.get_voxels(x2, *rig2)                                          # before b1's backward
        ((b1 * w1).sum() + (b2 * w2).sum()).backward()
        return [p.grad.clone() for p in params if p.grad is not None]
    ref = grads(True)
    got = grads(False)
    slots = m._plan_ws_slots
    assert len(slots) == 2 and slots[0]['ws'].data_ptr() != slots[1]['ws'].data_ptr()
    scale = max(float(g.abs().max()) for g in ref)
    assert scale > 0
    for a, b in zip(got, ref):
        assert float((a - b).abs().max()) <= 1e-6 * scale
    assert all(sl['plan']() is None for sl in slots)                          # the graph is gone: both slots are free again
    got2 = grads(False)
    assert len(m._plan_ws_slots) == 2
    for a, b in zip(got2, ref):
        assert float((a - b).abs().max()) <= 1e-6 * scale
