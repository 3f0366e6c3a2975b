"""Host-side logic of the terrain encoder (no GPU): grid constants, frustum and geometry vs the reference's golden
vectors; backbone parameter names / shapes (state_dict compatibility with reference checkpoints)."""
import numpy as np
import torch

from tests import helpers as hp

LSS_SMALL = dict(
    grid_conf=dict(xbound=[-3.2, 3.2, 0.2], ybound=[-3.2, 3.2, 0.2], zbound=[-2.0, 2.0, 2.0], dbound=[0.6, 3.4, 0.4]),
    data_aug_conf=dict(final_dim=(64, 96), H=64, W=96))


def test_grid_constants_and_frustum_match_reference():
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    g = hp.load('lss')
    m = LiftSplatShoot(LSS_SMALL['grid_conf'], LSS_SMALL['data_aug_conf'], build_backbones=False)
    assert np.array_equal(m.dx.numpy(), g['dx']) and np.array_equal(m.bx.numpy(), g['bx']) and np.array_equal(m.nx.numpy(), g['nx'])
    assert np.array_equal(m.frustum.numpy(), g['frustum'])
    assert m.nx.dtype == torch.int64 and not m.dx.requires_grad and 'frustum' in m.state_dict()


def test_get_geometry_matches_reference():
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    g = hp.load('lss')
    m = LiftSplatShoot(LSS_SMALL['grid_conf'], LSS_SMALL['data_aug_conf'], build_backbones=False)
    geom = m.get_geometry(*(torch.as_tensor(g[k]) for k in ('rots', 'trans', 'intrins', 'post_rots', 'post_trans'))).numpy()
    ref = g['geom'].copy()
    mask = np.ones(ref.shape[:-1], bool)
    mask[0, 0, 0, 0, :4] = False            # the generator overwrote these four points with hand-made edge cases
    assert np.array_equal(geom[mask], ref[mask])      # same torch ops in the same order -> bit-exact on CPU


def test_full_size_depth_bins():
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
    m = LiftSplatShoot(gc, dict(final_dim=(256, 512)), build_backbones=False)
    assert m.D == 59 and tuple(m.frustum.shape) == (59, 16, 32, 3) and m.nx.tolist() == [256, 256, 1]


def test_backbone_state_dict_names_and_shapes():
    """Keys a reference checkpoint carries, incl. the unused EfficientNet head."""
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    m = LiftSplatShoot(LSS_SMALL['grid_conf'], LSS_SMALL['data_aug_conf'], outC=1)
    sd = m.state_dict()
    expect = {
        'camencode.trunk._conv_stem.weight': (32, 3, 3, 3), 'camencode.trunk._bn0.running_mean': (32,),
        'camencode.trunk._blocks.0._depthwise_conv.weight': (32, 1, 3, 3), 'camencode.trunk._blocks.0._se_reduce.weight': (8, 32, 1, 1),
        'camencode.trunk._blocks.0._project_conv.weight': (16, 32, 1, 1), 'camencode.trunk._blocks.1._expand_conv.weight': (96, 16, 1, 1),
        'camencode.trunk._blocks.1._se_reduce.bias': (4,), 'camencode.trunk._blocks.3._depthwise_conv.weight': (144, 1, 5, 5),
        'camencode.trunk._blocks.10._project_conv.weight': (112, 672, 1, 1), 'camencode.trunk._blocks.15._project_conv.weight': (320, 1152, 1, 1),
        'camencode.trunk._conv_head.weight': (1280, 320, 1, 1), 'camencode.trunk._fc.weight': (1000, 1280),
        'camencode.up1.conv.0.weight': (512, 432, 3, 3), 'camencode.depthnet.weight': (7 + 64, 512, 1, 1), 'camencode.depthnet.bias': (71,),
        'bevencode.conv1.weight': (64, 64, 7, 7), 'bevencode.layer1.0.conv1.weight': (64, 64, 3, 3),
        'bevencode.layer2.0.downsample.0.weight': (128, 64, 1, 1), 'bevencode.layer3.1.bn2.weight': (256,),
        'bevencode.up1.conv.0.weight': (256, 320, 3, 3), 'bevencode.up_geom.1.weight': (128, 256, 3, 3),
        'bevencode.up_friction.4.weight': (1, 128, 1, 1), 'bevencode.up_diff.4.bias': (1,), 'dx': (3,), 'bx': (3,), 'nx': (3,),
    }
    for k, shp in expect.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape))
    assert '_expand_conv.weight' not in ' '.join(k for k in sd if k.startswith('camencode.trunk._blocks.0.'))
    assert len([k for k in sd if k.startswith('camencode.trunk._blocks.')]) > 0 and len(m.camencode.trunk._blocks) == 16
    n_params = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert 12e6 < n_params < 16e6          # SURVEY 8e estimate: ~13.6 M
    assert float(m.bevencode.layer1[0].bn2.weight.abs().sum()) == 0.0      # zero_init_residual


def test_encoder_runs_on_cpu_and_head_ranges():
    """The torch.nn parts are device-agnostic; only the splat needs the GPU.  Check lift shapes and head activations."""
    from monoforce_amd.terrain_encoder import CamEncode, BevEncode
    ce = CamEncode(D=7, C=64).eval()
    with torch.no_grad():
        lifted = ce(torch.randn(2, 3, 64, 96))
    assert tuple(lifted.shape) == (2, 64, 7, 4, 6)
    be = BevEncode(inC=64, outC=1).eval()
    with torch.no_grad():
        out = be(torch.randn(1, 64, 32, 32))
    assert set(out) == {'geom', 'terrain', 'diff', 'friction'} and tuple(out['geom'].shape) == (1, 1, 32, 32)
    assert float(out['geom'].abs().max()) <= 1.0 and float(out['diff'].min()) >= 0.0 and float(out['friction'].min()) >= 0.0
    assert torch.equal(out['terrain'], out['geom'] - out['diff'])


def test_points_from_obj_voxel_average(tmp_path):
    """`points_from_obj` (the stand-in for open3d's voxel_down_sample, dphys_config.py:26-31) on a hand-computed case: vertices
    are averaged per 0.1 m cube of the grid anchored at (min - voxel/2); faces / normals / comments are ignored."""
    from monoforce_amd.dphys_config import points_from_obj
    obj = tmp_path / 'body.obj'
    obj.write_text('\n'.join([
        '# test body', 'o body',
        'v 0.00 0.00 0.00', 'v 0.02 0.01 0.03',          # same cube (anchor -0.05): average (0.01, 0.005, 0.015)
        'v 0.30 0.00 0.00',                               # alone
        'v 0.30 0.26 0.00', 'v 0.34 0.28 0.04', 'v 0.32 0.27 0.02',   # same cube: average (0.32, 0.27, 0.02)
        'v 0.06 0.00 0.00',                               # next cube along x (boundary at 0.05)
        'vn 0 0 1', 'f 1 2 3', '']))
    pts = points_from_obj(str(obj), voxel_size=0.1)
    got = sorted(tuple(round(float(v), 6) for v in p) for p in pts)
    want = sorted([(0.01, 0.005, 0.015), (0.3, 0.0, 0.0), (0.32, 0.27, 0.02), (0.06, 0.0, 0.0)])
    assert len(got) == 4 and all(max(abs(a - b) for a, b in zip(g, w)) < 1e-6 for g, w in zip(got, want)), got
    raw = points_from_obj(str(obj), voxel_size=0)
    assert tuple(raw.shape) == (7, 3)


def test_missing_mesh_warns_about_the_standin(monkeypatch):
    import warnings
    from monoforce_amd.dphys_config import DPhysConfig
    monkeypatch.delenv('MONOFORCE_MESH_DIR', raising=False)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        cfg = DPhysConfig(robot='tradr')
    assert any('STAND-IN' in str(w.message) for w in rec) and cfg.robot_points.shape[1] == 3


def test_backbone_launch_diet_is_the_same_arithmetic(monkeypatch):
    """MF_BACKBONE_LEAN (symmetric pad inside the convolution, squeeze-excitation as two small GEMMs, drop-connect + residual as
    one addcmul) against the plain module graph: same draws, same values and gradients."""
    from monoforce_amd import backbones as bb

    def run(flag):
        monkeypatch.setenv('MF_BACKBONE_LEAN', flag)
        torch.manual_seed(1)
        blocks = [bb.MBConvBlock(16, 16, 3, 1, 6), bb.MBConvBlock(16, 24, 5, 2, 6), bb.MBConvBlock(24, 24, 5, 1, 6)]
        torch.manual_seed(2)
        x = torch.randn(6, 16, 12, 10, requires_grad=True)
        y = x
        for b in blocks:
            y = b(y, 0.3)
        y.square().sum().backward()
        return [y.detach(), x.grad] + [p.grad for b in blocks for p in b.parameters()]

    for a, b in zip(run('1'), run('0')):
        assert float(b.abs().max()) > 0 and torch.allclose(a, b, rtol=1e-5, atol=2e-6 * float(b.abs().max()))


def test_batchnorm_counters_count_once_per_forward_and_survive_moves():
    import copy
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    m = LiftSplatShoot(LSS_SMALL['grid_conf'], LSS_SMALL['data_aug_conf'])
    g = hp.load('lss')
    cal = [torch.as_tensor(g[k]) for k in ('rots', 'trans', 'intrins', 'post_rots', 'post_trans')]
    x = torch.randn(cal[0].shape[0], cal[0].shape[1], 3, 64, 96)

    def voxels_without_the_splat(self, x, *cal, plan=None):     # the splat needs the GPU; every layer still runs once
        B, N, C, H, W = x.shape
        self.camencode.get_depth_and_context(x.view(B * N, C, H, W))
        return torch.randn(B, self.camC, int(self.nx[0]), int(self.nx[1]), dtype=x.dtype)

    LiftSplatShoot.get_voxels, keep = voxels_without_the_splat, LiftSplatShoot.get_voxels
    try:
        _counter_checks(m, x, cal)
    finally:
        LiftSplatShoot.get_voxels = keep


def _counter_checks(m, x, cal):
    import copy
    m.train()
    for _ in range(2):
        m(x, *cal)
    def counts(mod):      # every batch norm that RUNS; the EfficientNet head's `_bn1` never does (lss.py:78-92) and stays at 0, like the reference's
        sd = mod.state_dict()
        idle = [k for k in sd if k.endswith('trunk._bn1.num_batches_tracked')]
        assert len(idle) == 1 and int(sd[idle[0]]) == 0, idle
        return {k: int(v) for k, v in sd.items() if k.endswith('num_batches_tracked') and k not in idle}
    counters = counts(m)
    assert len(counters) > 60 and set(counters.values()) == {2}
    m2 = copy.deepcopy(m).double()            # buffers re-created one by one: the bank gathers them again
    m2(x.double(), *(c.double() for c in cal))
    assert set(counts(m2).values()) == {3}
    assert set(counts(m).values()) == {2}
    m.eval()
    m(x, *cal)
    assert set(counts(m).values()) == {2}
    m.load_state_dict(m2.float().state_dict())
    assert int(m.camencode.trunk._bn0.num_batches_tracked) == 3


def test_batchnorm_counters_follow_direct_submodule_calls_and_the_model_pickles(tmp_path):
    """a training forward that calls the sub-modules directly (the fused lift, the reference's notebooks) advances the
    counters of the batch norms that ran -- and only those; the counted batch norm is a class, so `torch.save(model)` works."""
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    from monoforce_amd.backbones import BatchNorm2dCounted
    m = LiftSplatShoot(LSS_SMALL['grid_conf'], LSS_SMALL['data_aug_conf']).train()
    x = torch.randn(2, 3, 64, 96)
    m.camencode.get_depth_and_context(x)
    m.camencode(x)
    cam = {int(v) for k, v in m.state_dict().items() if k.endswith('num_batches_tracked') and k.startswith('camencode') and 'trunk._bn1.' not in k}
    bev = {int(v) for k, v in m.state_dict().items() if k.endswith('num_batches_tracked') and k.startswith('bevencode')}
    assert cam == {2} and bev == {0} and int(m.camencode.trunk._bn1.num_batches_tracked) == 0
    m.bevencode(torch.randn(1, m.camC, int(m.nx[0]), int(m.nx[1])))
    assert {int(v) for k, v in m.state_dict().items() if k.endswith('num_batches_tracked') and k.startswith('bevencode')} == {1}
    assert sum(isinstance(b, BatchNorm2dCounted) for b in m.modules()) > 60
    path = tmp_path / 'model.pt'
    torch.save(m, path)
    m2 = torch.load(path, weights_only=False)
    m2.train().camencode(x)
    assert int(m2.camencode.trunk._bn0.num_batches_tracked) == 3 and int(m.camencode.trunk._bn0.num_batches_tracked) == 2


def test_batchnorm_counters_after_a_syncbatchnorm_conversion():
    """`SyncBatchNorm.convert_sync_batchnorm` replaces the counted layers; the replacements share the old buffer tensors and count
    for themselves.  The bank drops what it no longer owns -- a converted layer advanced by TWO per step before -- and keeps bumping the rest."""
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    from monoforce_amd.backbones import bump_batchnorm_counters, BatchNorm2dCounted
    m = LiftSplatShoot(LSS_SMALL['grid_conf'], LSS_SMALL['data_aug_conf']).train()
    bump_batchnorm_counters(m.camencode); bump_batchnorm_counters(m.bevencode)
    assert int(m.camencode.trunk._bn0.num_batches_tracked) == 1 and int(m.camencode.trunk._bn1.num_batches_tracked) == 0
    m.camencode = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m.camencode)          # (the bank object stays on the same owner module)
    sync = [b for b in m.camencode.modules() if isinstance(b, torch.nn.SyncBatchNorm)]
    assert len(sync) > 40 and not any(isinstance(b, BatchNorm2dCounted) for b in m.camencode.modules())
    before = [int(b.num_batches_tracked) for b in sync]
    bump_batchnorm_counters(m.camencode); bump_batchnorm_counters(m.bevencode)
    assert [int(b.num_batches_tracked) for b in sync] == before                        # theirs to bump now (their own forward does it)
    assert {int(v) for k, v in m.state_dict().items() if k.endswith('num_batches_tracked') and k.startswith('bevencode')} == {2}
