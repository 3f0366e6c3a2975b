"""The `monoforce.*` import surface the reference's scripts use resolves to the MI355X implementation."""


def test_reference_import_paths():
    from monoforce.models.traj_predictor.dphys_config import DPhysConfig
    from monoforce.models.traj_predictor.dphysics import DPhysics, generate_controls
    from monoforce.models.terrain_encoder.lss import LiftSplatShoot
    from monoforce.models.terrain_encoder.utils import gen_dx_bx, QuickCumsum, cumsum_trick  # noqa: F401
    from monoforce.losses import physics_loss, hm_loss, total_variation  # noqa: F401
    from monoforce.utils import read_yaml, write_to_yaml, str2bool  # noqa: F401
    import monoforce_amd.dphysics as impl
    assert DPhysics is impl.DPhysics and LiftSplatShoot.__module__ == 'monoforce_amd.terrain_encoder'
    cfg = DPhysConfig(robot='tradr')             # no mesh in the image -> documented stand-in body
    assert cfg.robot_mass == 40.0 and cfg.robot_points.shape[1] == 3 and len(cfg.driving_parts) == 2
    assert cfg.use_odeint and cfg.dt == 0.01 and cfg.traj_sim_time == 5.0 and cfg.d_max == 6.4
    dp = DPhysics(cfg, device='cpu')             # construction is device-free; forward needs the GPU
    assert dp.ts.shape[0] == 500 and dp.I_inv.shape == (1, 3, 3)
    c, ts = generate_controls(n_trajs=3, time_horizon=5.0, dt=0.01, v_range=(0.5, 1.0), w_range=(-1, 1))
    assert tuple(c.shape) == (3, 500, 2) and ts.shape[0] == 500 and bool((c[:, 0] == c[:, -1]).all())


def test_cumsum_trick_and_quickcumsum_agree():
    import torch
    from monoforce.models.terrain_encoder.utils import QuickCumsum, cumsum_trick
    x = torch.randn(10, 3, dtype=torch.float64, requires_grad=True)
    ranks = torch.tensor([0, 0, 1, 3, 3, 3, 4, 7, 7, 9])
    geom = torch.arange(10).unsqueeze(1)
    a, ga = cumsum_trick(x, geom, ranks)
    b, gb = QuickCumsum.apply(x, geom, ranks)
    assert torch.allclose(a, b) and torch.equal(ga, gb)
    ref = torch.stack([x[ranks == r].sum(0) for r in ranks.unique()])
    assert torch.allclose(a, ref)
    (b * torch.arange(6, dtype=torch.float64).unsqueeze(1)).sum().backward()
    seg = torch.tensor([0, 0, 1, 2, 2, 2, 3, 4, 4, 5], dtype=torch.float64)
    assert torch.allclose(x.grad, seg.unsqueeze(1).expand(-1, 3))


def test_yaml_roundtrip(tmp_path):
    from monoforce.models.traj_predictor.dphys_config import DPhysConfig
    cfg = DPhysConfig(robot='husky')
    p = str(tmp_path / 'cfg.yaml')
    cfg.to_yaml(p)
    cfg2 = DPhysConfig(robot='husky')
    cfg2.from_yaml(p)
    assert cfg2.robot_mass == 50.0 and cfg2.grid_res == cfg.grid_res and isinstance(cfg2.robot_points, list)
