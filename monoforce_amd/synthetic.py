"""Deterministic synthetic inputs for the DPhysics rollout / LSS splat hot path.

Everything here is a closed-form function of a few stored parameters, evaluated in float64 with
numpy and only then cast, so the golden-vector generator (run once in the build container) and
the tests / bench (run on the GPU box) regenerate bit-identical inputs without shipping big maps.

Shapes follow the reference: height / friction grids are ``[B, H, W]`` with the FIRST grid axis = x
(`/root/reference/monoforce/src/monoforce/models/traj_predictor/dphysics.py:427-430`), controls are
``[B, T, 2] = (v, w)`` constant in time per rollout (`dphysics.py:62-70`).
"""
import numpy as np
import torch

__all__ = ['robot_points_4', 'robot_points_box', 'bump_terrain', 'wave_friction', 'const_controls',
           'lss_camera_rig']


def robot_points_4():
    """4 contact points (+-0.25, +-0.27, -0.1) m, two tracks: left = y > 0, right = y < 0."""
    pts = np.array([[0.25, 0.27, -0.1], [0.25, -0.27, -0.1], [-0.25, 0.27, -0.1], [-0.25, -0.27, -0.1]], np.float64)
    left = pts[:, 1] > 0
    return pts.astype(np.float32), [left, ~left]


def robot_points_box(n, seed=0, n_tracks=2):
    """n pseudo-random body points in a tradr-sized box; lower outer points are the driving parts.

    Masks follow the split rules of `dphys_config.py:46-63` (2 tracks: |y - cog| > s_y/4 and z < cog_z;
    4 tracks: |x - cog| > s_x/8 and |y - cog| > s_y/3) so that some points are NOT driving.
    """
    rng = np.random.RandomState(seed)
    pts = (rng.rand(n, 3) - 0.5) * np.array([0.8, 0.6, 0.25]) + np.array([0.0, 0.0, -0.02])
    pts = pts.astype(np.float32)
    cog = pts.mean(0)
    sx, sy = pts[:, 0].max() - pts[:, 0].min(), pts[:, 1].max() - pts[:, 1].min()
    if n_tracks == 2:
        masks = [(pts[:, 1] > cog[1] + sy / 4) & (pts[:, 2] < cog[2]),
                 (pts[:, 1] < cog[1] - sy / 4) & (pts[:, 2] < cog[2])]
    else:
        masks = [(pts[:, 0] > cog[0] + sx / 8) & (pts[:, 1] > cog[1] + sy / 3),
                 (pts[:, 0] > cog[0] + sx / 8) & (pts[:, 1] < cog[1] - sy / 3),
                 (pts[:, 0] < cog[0] - sx / 8) & (pts[:, 1] > cog[1] + sy / 3),
                 (pts[:, 0] < cog[0] - sx / 8) & (pts[:, 1] < cog[1] - sy / 3)]
    return pts, masks


def _grid_xy(d_max, grid_res):
    # same node positions as dphys_config.py:137-139 (arange(-d_max, d_max, res), indexing='ij')
    n = int(round(2 * d_max / grid_res))
    ax = -d_max + grid_res * np.arange(n, dtype=np.float64)
    return np.meshgrid(ax, ax, indexing='ij')


def bump_params(seed, n_bumps=6, amp=0.6, smooth=False):
    """Parameters of a sum-of-Gaussians terrain (cf. examples/diff_physics.ipynb, fit_terrain.py:26)."""
    rng = np.random.RandomState(seed)
    if smooth:  # single wide bump ahead of the robot (robot_control.py:101 style)
        return np.array([[0.25, 2.0, 0.0, 4.0]], np.float64)
    a = rng.rand(n_bumps) * amp
    c = rng.rand(n_bumps, 2) * 10.0 - 5.0
    s = 0.5 + 2.0 * rng.rand(n_bumps)
    return np.concatenate([a[:, None], c, s[:, None]], 1)  # [n, (amp, cx, cy, width)]


def bump_terrain(params, d_max=6.4, grid_res=0.05, dtype=torch.float32):
    """z[H, W] = sum_k amp_k * exp(-((x-cx_k)^2 + (y-cy_k)^2) / s_k)."""
    X, Y = _grid_xy(d_max, grid_res)
    z = np.zeros_like(X)
    for a, cx, cy, s in np.asarray(params, np.float64):
        z += a * np.exp(-((X - cx) ** 2 + (Y - cy) ** 2) / s)
    return torch.from_numpy(z).to(dtype)


def wave_friction(d_max=6.4, grid_res=0.05, lo=0.5, hi=1.0, kx=1.3, ky=0.9, dtype=torch.float32):
    """Smooth friction map in [lo, hi]: mid + half * sin(kx x) * cos(ky y)."""
    X, Y = _grid_xy(d_max, grid_res)
    mu = 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.sin(kx * X) * np.cos(ky * Y)
    return torch.from_numpy(mu).to(dtype)


def const_controls(B, T, seed=0, v_range=(0.5, 1.0), w_range=(-2.0, 2.0), dtype=torch.float32):
    """Constant-in-time (v, w) per rollout, `generate_controls` semantics (dphysics.py:62-70)."""
    rng = np.random.RandomState(seed)
    v = v_range[0] + (v_range[1] - v_range[0]) * rng.rand(B)
    w = w_range[0] + (w_range[1] - w_range[0]) * rng.rand(B)
    c = np.stack([np.repeat(v[:, None], T, 1), np.repeat(w[:, None], T, 1)], -1)
    return torch.from_numpy(c).to(dtype)


def varying_controls(B, T, seed=0, dtype=torch.float32):
    """Time-varying controls (exercise the per-step control lookup, dphysics.py:183-184)."""
    rng = np.random.RandomState(seed)
    t = np.linspace(0.0, 1.0, T)[None, :]
    v = 0.75 + 0.25 * np.sin(2 * np.pi * (t * rng.uniform(0.5, 2.0, (B, 1)) + rng.rand(B, 1)))
    w = 1.5 * np.sin(2 * np.pi * (t * rng.uniform(0.5, 3.0, (B, 1)) + rng.rand(B, 1)))
    return torch.from_numpy(np.stack([v, w], -1)).to(dtype)


def lss_camera_rig(B, n_cams=4, H=256, W=512, f=300.0, dtype=torch.float32):
    """n_cams pinhole cameras yawed 360/n_cams deg apart at t=(0.3, 0, 0.5) rotated with the yaw; identity aug.

    Camera frame is x-right / y-down / z-forward; ego frame x-forward / y-left / z-up.
    Returns rots[B,n,3,3], trans[B,n,3], intrins[B,n,3,3], post_rots[B,n,3,3], post_trans[B,n,3].
    """
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float64)
    cam2ego0 = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)
    rots, trans = [], []
    for i in range(n_cams):
        a = 2 * np.pi * i / n_cams
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        rots.append(Rz @ cam2ego0)
        trans.append(Rz @ np.array([0.3, 0.0, 0.5]))
    rots = torch.from_numpy(np.stack(rots)).to(dtype).expand(B, -1, -1, -1).contiguous()
    trans = torch.from_numpy(np.stack(trans)).to(dtype).expand(B, -1, -1).contiguous()
    intrins = torch.from_numpy(K).to(dtype).expand(B, n_cams, -1, -1).contiguous()
    post_rots = torch.eye(3, dtype=dtype).expand(B, n_cams, -1, -1).contiguous()
    post_trans = torch.zeros(B, n_cams, 3, dtype=dtype)
    return rots, trans, intrins, post_rots, post_trans


def probe_weights(shape, phase=0.0, dtype=torch.float32):
    """Closed-form pseudo-random weights sin(0.37 i + phase) used to build scalar test losses from tensor outputs."""
    n = int(np.prod(shape))
    w = np.sin(0.37 * np.arange(n, dtype=np.float64) + phase).reshape(tuple(shape))
    return torch.from_numpy(w).to(dtype)
