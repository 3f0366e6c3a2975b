"""`estimate_heightmap` of the reference's label generation (`/root/reference/monoforce/src/monoforce/cloudproc.py:88-148`)
on the MI355X: per-cell maximum height of a point cloud plus the measurement mask, `mf_estimate_heightmap_f32`
(csrc/heightmap.hip; include/monoforce_hip.h).  Same signature, same `[2, H, W]` float32 result (height, mask; first grid
axis = x).  The cloud may live on the host (it is copied to the GPU, the result comes back on the cloud's device) or on the
GPU; there is no CPU implementation.
"""
import ctypes as C

import torch

from . import _lib

__all__ = ['estimate_heightmap']

_BINS = {}


def _bin_edges(d_max, grid_res, device):
    """The reference's bin edges: `torch.arange(-d_max, d_max, grid_res)` in the default dtype (float32), built on the host
    exactly like there -- the edges, not a formula, decide which cell a point on a boundary falls into."""
    key = (float(d_max), float(grid_res), str(device))
    if key not in _BINS:
        _BINS[key] = torch.arange(-d_max, d_max, grid_res, dtype=torch.float32).to(device)
    return _BINS[key]


def estimate_heightmap(points, grid_res, d_max, h_max, r_min=None, h_min=None):
    """points [N, 3] (x, y, z) float32 -> hm [2, H, W]: hm[0] = max z per cell (0 where no point fell), hm[1] = 1.0 where
    measured.  Rows with NaNs, points closer than `r_min` to the origin (xy) and points outside the open box
    (-d_max, d_max)^2 x (h_min, h_max) are ignored; `h_min` defaults to -h_max."""
    assert points.dim() == 2 and points.shape[1] >= 3
    if not torch.cuda.is_available():
        raise RuntimeError('estimate_heightmap runs on the MI355X HIP path only (no CPU fallback)')
    home = points.device
    dev = home if points.is_cuda else torch.device('cuda', torch.cuda.current_device())
    pts = points.detach()[:, :3].to(device=dev, dtype=torch.float32).contiguous()
    if points.shape[1] > 3:
        # the reference drops a row when ANY of its columns is NaN (cloudproc.py:90: `~torch.isnan(points).any(dim=1)`), the extra
        # fields (intensity, ...) included: such a row gets a NaN x here, which the kernel's own filter then discards
        extra_nan = torch.isnan(points.detach()[:, 3:]).any(dim=1).to(dev)
        pts[:, 0] = torch.where(extra_nan, torch.full_like(pts[:, 0], float('nan')), pts[:, 0])
    xb = _bin_edges(d_max, grid_res, dev)
    n = xb.numel()
    desc = _lib.MfHeightmapDesc(n_points=pts.shape[0], nx=n, ny=n, d_max=float(d_max), h_min=float(-h_max if h_min is None else h_min),
                                h_max=float(h_max), r_min=float(-1.0 if r_min is None else r_min), inv_res=float(1.0 / grid_res))
    scratch = torch.empty(n * n, dtype=torch.int32, device=dev)
    hm = torch.empty(2, n, n, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mf_estimate_heightmap_f32(C.byref(desc), _lib.ptr(pts), _lib.ptr(xb), _lib.ptr(xb), _lib.ptr(scratch),
                                                        _lib.ptr(hm), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   'mf_estimate_heightmap')
    return hm.to(home)
