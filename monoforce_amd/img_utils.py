"""Host-side image / camera helpers of the terrain encoder under the reference's names (SURVEY.md 8f row 4;
`/root/reference/monoforce/src/monoforce/models/terrain_encoder/utils.py:13-133`: `ego_to_cam` :13, `cam_to_ego` :25,
`get_only_in_img_mask` :38, `get_rot` :46, `img_transform` :52, `NormalizeInverse` :82, `denormalize_img` :96,
`normalize_img` :102, `resize_img` :107, `sample_augmentation` :110).

These run once per camera frame on the host (PIL images, 3x3 matrices) before anything reaches the GPU; they are here so
that `scripts/run.py` / `scripts/train.py` find every name they import.  Own implementation: the augmentation is composed as
ONE 2-D affine map (A, b) that is applied to (post_rot, post_tran) at the end, the normalisation transforms are small
callables on torch tensors (torchvision is not a dependency), same call signatures and the same numbers.
"""
import numpy as np
import torch
from PIL import Image

__all__ = ['ego_to_cam', 'cam_to_ego', 'get_only_in_img_mask', 'get_rot', 'img_transform', 'NormalizeInverse',
           'denormalize_img', 'normalize_img', 'resize_img', 'sample_augmentation', 'mean', 'std']

# ImageNet statistics the backbone was trained with
mean = [0.485, 0.456, 0.406]
std = [0.229, 0.224, 0.225]


def ego_to_cam(points, rot, trans, intrins):
    """Ego-frame points [3, N] -> pixel coordinates (u, v) and depth of a pinhole camera with pose (rot, trans)."""
    cam = rot.transpose(0, 1).matmul(points - trans.unsqueeze(1))      # R^T (p - t)
    uvw = intrins.matmul(cam)
    uvw[:2] /= uvw[2:3]                                                 # in place, like the reference (a fresh tensor anyway)
    return uvw


def cam_to_ego(points, rot, trans, intrins):
    """Inverse of `ego_to_cam`: (u, v, depth) [3, N] -> ego-frame points."""
    depth = points[2:3]
    rays = torch.cat((points[:2] * depth, depth))
    out = rot.matmul(intrins.inverse().matmul(rays))
    out += trans.unsqueeze(1)
    return out


def get_only_in_img_mask(pts, H, W):
    """True for projected points [3, N] in front of the camera and more than one pixel inside the H x W image."""
    u, v, d = pts[0], pts[1], pts[2]
    return (d > 0) & (u > 1) & (u < W - 1) & (v > 1) & (v < H - 1)


def get_rot(h):
    """2-D rotation by -h (the image rotates by +h, its pixel coordinates by -h); float32 like `torch.Tensor`."""
    c, s = np.cos(h), np.sin(h)
    return torch.Tensor([[c, s], [-s, c]])


def img_transform(img, post_rot, post_tran, resize, resize_dims, crop, flip, rotate):
    """Resize, crop, optionally mirror and rotate a PIL image, and push the same pixel map onto (post_rot, post_tran) so
    that `p_aug = post_rot @ p + post_tran` keeps relating original to augmented pixel coordinates.

    `post_rot` / `post_tran` are scaled / shifted IN PLACE for the resize and crop (the reference's `*=` / `-=`), and the
    mirror + rotation part is returned as new tensors -- callers use the return values."""
    img = img.resize(resize_dims).crop(crop)
    if flip:
        img = img.transpose(method=Image.FLIP_LEFT_RIGHT)
    img = img.rotate(rotate)

    post_rot *= resize
    post_tran -= torch.Tensor(crop[:2])
    w, h = crop[2] - crop[0], crop[3] - crop[1]
    if flip:                                   # u -> w - u
        mirror = torch.Tensor([[-1, 0], [0, 1]])
        post_rot = mirror.matmul(post_rot)
        post_tran = mirror.matmul(post_tran) + torch.Tensor([w, 0])
    # rotation about the centre of the cropped image: p -> A (p - c) + c
    A = get_rot(rotate / 180 * np.pi)
    centre = torch.Tensor([w, h]) / 2
    shift = A.matmul(-centre) + centre
    return img, A.matmul(post_rot), A.matmul(post_tran) + shift


class _Normalize:
    """(x - mean) / std per channel of a [C, H, W] float tensor (torchvision.transforms.Normalize semantics)."""

    def __init__(self, mean, std):
        self.mean = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, tensor):
        return (tensor - self.mean.to(tensor)) / self.std.to(tensor)


class NormalizeInverse(_Normalize):
    """Undo `_Normalize(mean, std)`: x * std + mean, expressed as a normalisation with (-mean / std, 1 / std)."""

    def __init__(self, mean, std):
        m, s = torch.as_tensor(mean, dtype=torch.float32), torch.as_tensor(std, dtype=torch.float32)
        s_inv = 1 / (s + 1e-7)
        super().__init__(mean=-m * s_inv, std=s_inv)

    def __call__(self, tensor):
        return super().__call__(tensor.clone())


def _to_tensor(pic):
    """PIL image / HxWxC uint8 array -> float32 [C, H, W] in [0, 1] (torchvision ToTensor semantics)."""
    arr = np.asarray(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


def _to_pil(tensor):
    """float [C, H, W] in [0, 1] -> PIL image (torchvision ToPILImage semantics: x * 255, truncated to uint8)."""
    arr = tensor.detach().cpu().mul(255).to(torch.uint8).permute(1, 2, 0).numpy()
    return Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr)


class _Compose:
    def __init__(self, *fns):
        self.fns = fns

    def __call__(self, x):
        for f in self.fns:
            x = f(x)
        return x


class _ResizeShortSide:
    """Bicubic resize so that the shorter image side becomes `size` (torchvision Resize(int) semantics)."""

    def __init__(self, size, interpolation=Image.BICUBIC):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img):
        w, h = img.size
        if w <= h:
            new = (self.size, int(self.size * h / w))
        else:
            new = (int(self.size * w / h), self.size)
        return img.resize(new, self.interpolation)


normalize_img = _Compose(_to_tensor, _Normalize(mean=mean, std=std))
denormalize_img = _Compose(NormalizeInverse(mean=mean, std=std), _to_pil)
resize_img = _ResizeShortSide(512, interpolation=Image.BICUBIC)


def sample_augmentation(lss_cfg, is_train=False):
    """(resize, resize_dims, crop, flip, rotate) for one camera image: random within the configured limits for training,
    the centred deterministic variant otherwise.  Draws from `np.random` in the reference's order (resize, bottom
    fraction, horizontal offset, flip, rotation) so that seeded runs reproduce."""
    aug = lss_cfg['data_aug_conf']
    H, W = aug['H'], aug['W']
    fH, fW = aug['final_dim']
    if is_train:
        resize = np.random.uniform(*aug['resize_lim'])
        newW, newH = int(W * resize), int(H * resize)
        crop_h = int((1 - np.random.uniform(*aug['bot_pct_lim'])) * newH) - fH
        crop_w = int(np.random.uniform(0, max(0, newW - fW)))
        flip = bool(aug['rand_flip'] and np.random.choice([0, 1]))
        rotate = np.random.uniform(*aug['rot_lim'])
    else:
        resize = max(fH / H, fW / W)
        newW, newH = int(W * resize), int(H * resize)
        crop_h = int((1 - np.mean(aug['bot_pct_lim'])) * newH) - fH
        crop_w = int(max(0, newW - fW) / 2)
        flip, rotate = False, 0
    return resize, (newW, newH), (crop_w, crop_h, crop_w + fW, crop_h + fH), flip, rotate
