"""Host-side image / camera helpers vs vectors produced by the reference's own functions
(tests/golden/gen_golden.py::gen_img_utils), and the import surface of the reference's scripts."""
import numpy as np
import torch
from PIL import Image

from tests import helpers as hp
from tests.golden_images import test_image


def test_projection_helpers_match_reference():
    from monoforce_amd import img_utils as U
    g = hp.load('img_utils')
    pts, rot, trans, K = (torch.as_tensor(g['proj/' + k]) for k in ('pts', 'rot', 'trans', 'K'))
    cam = U.ego_to_cam(pts.clone(), rot, trans, K)
    assert np.array_equal(cam.numpy(), g['proj/cam'])
    assert np.array_equal(U.get_only_in_img_mask(cam, 256, 512).numpy(), g['proj/mask']) and 0 < g['proj/mask'].sum() < 40
    assert np.array_equal(U.cam_to_ego(cam.clone(), rot, trans, K).numpy(), g['proj/back'])
    assert np.abs(g['proj/back'] - g['proj/pts']).max() < 1e-4
    got = np.stack([U.get_rot(h).numpy() for h in (0.0, 0.1, -0.7, np.pi / 2)])
    assert np.array_equal(got, g['get_rot'])


def _pack(t):
    resize, dims, crop, flip, rotate = t
    return np.array([resize, dims[0], dims[1], crop[0], crop[1], crop[2], crop[3], float(flip), rotate], np.float64)


def test_sample_augmentation_matches_reference_draws():
    from monoforce_amd.img_utils import sample_augmentation
    g = hp.load('img_utils')
    cfg = dict(data_aug_conf=dict(H=1200, W=1920, final_dim=(256, 512), resize_lim=(0.25, 0.35), bot_pct_lim=(0.0, 0.1),
                                  rot_lim=(-5.4, 5.4), rand_flip=True))
    assert np.array_equal(_pack(sample_augmentation(cfg, is_train=False)), g['aug/eval'])
    np.random.seed(11)
    got = np.stack([_pack(sample_augmentation(cfg, is_train=True)) for _ in range(8)])
    assert np.array_equal(got, g['aug/train']) and 0 < got[:, 7].sum() < 8      # some flipped, some not


def test_img_transform_matches_reference():
    from monoforce_amd.img_utils import img_transform
    g = hp.load('img_utils')
    src = test_image(320, 200)
    for i in range(3):
        a = g[f'tf{i}/args']
        resize, dims, crop, flip, rotate = float(a[0]), (int(a[1]), int(a[2])), tuple(int(v) for v in a[3:7]), bool(a[7]), float(a[8])
        pr0, pt0 = torch.eye(2), torch.zeros(2)
        img, pr, pt = img_transform(Image.fromarray(src), pr0, pt0, resize=resize, resize_dims=dims, crop=crop, flip=flip, rotate=rotate)
        assert np.array_equal(np.asarray(img), g[f'tf{i}/img'])
        assert np.array_equal(pr.numpy(), g[f'tf{i}/post_rot']) and np.array_equal(pt.numpy(), g[f'tf{i}/post_tran'])
        # the reference's in-place side effects on the caller's tensors (resize / crop part)
        assert torch.equal(pr0, torch.eye(2) * resize) and torch.equal(pt0, -torch.Tensor(crop[:2]))


def test_normalisation_transforms_roundtrip():
    """torchvision is absent here and in the reference image of this build, so these follow torchvision's documented
    semantics (ToTensor: uint8 / 255, CHW; Normalize: (x - mean) / std; ToPILImage: x * 255 truncated) -- parity unpinned."""
    from monoforce_amd.img_utils import denormalize_img, normalize_img, resize_img, mean, std
    src = test_image(64, 48)
    t = normalize_img(Image.fromarray(src))
    assert t.shape == (3, 48, 64) and t.dtype == torch.float32
    ref = (torch.from_numpy(src).permute(2, 0, 1).float() / 255 - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
    assert torch.allclose(t, ref, atol=1e-6)
    back = np.asarray(denormalize_img(t))
    assert back.shape == src.shape and np.abs(back.astype(int) - src.astype(int)).max() <= 1
    big = resize_img(Image.fromarray(src))
    assert big.size == (int(512 * 64 / 48), 512)


def test_the_scripts_import_lines_resolve():
    """The `from monoforce... import ...` lines of the reference's scripts/run.py:11-15 and scripts/train.py:9-15, minus
    `monoforce.datasets` (ROUGH) and `monoforce.vis` (mayavi) which are out of scope."""
    from monoforce.models.traj_predictor.dphys_config import DPhysConfig  # noqa: F401
    from monoforce.models.traj_predictor.dphysics import DPhysics, generate_controls  # noqa: F401
    from monoforce.models.terrain_encoder.lss import LiftSplatShoot  # noqa: F401
    from monoforce.models.terrain_encoder.utils import denormalize_img, normalize_img, img_transform, sample_augmentation  # noqa: F401
    from monoforce.utils import read_yaml, load_calib  # noqa: F401
    from monoforce.models.terrain_encoder.utils import denormalize_img, ego_to_cam, get_only_in_img_mask  # noqa: F401,F811
    from monoforce.utils import read_yaml, write_to_yaml, str2bool, compile_data  # noqa: F401,F811
    from monoforce.losses import hm_loss, physics_loss  # noqa: F401
    from monoforce.cloudproc import estimate_heightmap  # noqa: F401
    import pytest
    with pytest.raises(NotImplementedError):
        compile_data()


def test_load_calib(tmp_path):
    import yaml
    from monoforce.utils import load_calib
    assert load_calib(str(tmp_path)) is None
    (tmp_path / 'cameras').mkdir()
    (tmp_path / 'cameras' / 'camera_front.yaml').write_text(yaml.dump({'image_width': 1920, 'camera_matrix': {'data': list(range(9))}}))
    T = np.eye(4); T[2, 3] = -0.132
    (tmp_path / 'transformations.yaml').write_text(yaml.dump({'T_base_link__base_footprint': {'data': T.flatten().tolist()}}))
    c = load_calib(str(tmp_path))
    assert set(c) == {'camera_front', 'transformations', 'clearance'} and abs(c['clearance'] - 0.132) < 1e-6
