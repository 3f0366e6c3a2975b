"""Terrain staging between the BEV heads and the rollout on `mf_terrain_stage_*`.

The reference does it in four steps spread over two files: `terrain = geom - diff` (lss.py:158), then
`AvgPool2d(k, k)` of `terrain` and of `friction` onto the coarser physics grid (scripts/train.py:93-99, 233-235), then the
rollout reads the two pooled maps.  `stage_terrain` is those steps as ONE kernel (and one in the backward): it returns the
full-resolution `terrain` (the height-map loss needs it), the pooled `z` and `mu` -- and leaves the interleaved `(z, mu)`
pair the rollout kernels gather from attached to `z`, where `DPhysics` finds it and skips its own interleave pass.
"""
import ctypes as C

import torch

from . import _lib, _timing

__all__ = ['stage_terrain', 'staged_pair']


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _Stage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geom, diff, friction, k):
        B, H, W = geom.shape
        dev = geom.device
        h, w = H // k, W // k
        terrain = torch.empty_like(geom)
        z = torch.empty(B, h, w, dtype=torch.float32, device=dev)
        mu = torch.empty_like(z)
        zmu = torch.empty(B, h, w, 2, dtype=torch.float32, device=dev)
        desc = _lib.MfStageDesc(B=B, H=H, W=W, k=k)
        with torch.cuda.device(dev), _timing.timed('terrain_stage_fwd', dev):
            _lib.check(_lib.lib().mf_terrain_stage_fwd_f32(C.byref(desc), _lib.ptr(geom), _lib.ptr(diff), _lib.ptr(friction), _lib.ptr(terrain),
                                                           _lib.ptr(z), _lib.ptr(mu), _lib.ptr(zmu), _stream_ptr(dev)), 'mf_terrain_stage_fwd')
        ctx.desc = desc
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(zmu)
        return terrain, z, mu, zmu

    @staticmethod
    def backward(ctx, g_terrain, gz, gmu, _g_zmu):
        desc = ctx.desc
        cont = lambda t: None if t is None else t.contiguous()  # noqa: E731
        g_terrain, gz, gmu = cont(g_terrain), cont(gz), cont(gmu)
        ref = next(t for t in (g_terrain, gz, gmu) if t is not None)
        dev = ref.device
        g_geom = torch.empty(desc.B, desc.H, desc.W, dtype=torch.float32, device=dev)
        g_diff, g_fric = torch.empty_like(g_geom), torch.empty_like(g_geom)
        with torch.cuda.device(dev), _timing.timed('terrain_stage_bwd', dev):
            _lib.check(_lib.lib().mf_terrain_stage_bwd_f32(C.byref(desc), _lib.ptr(g_terrain), _lib.ptr(gz), _lib.ptr(gmu), _lib.ptr(g_geom),
                                                           _lib.ptr(g_diff), _lib.ptr(g_fric), _stream_ptr(dev)), 'mf_terrain_stage_bwd')
        return g_geom, g_diff, g_fric, None


def stage_terrain(geom, diff, friction, k=1):
    """geom, diff, friction: float32 head outputs [B,1,H,W] or [B,H,W] on the GPU; k: pooling factor (physics grid_res / BEV grid_res).

    Returns (terrain [like geom], z [B,h,w], mu [B,h,w]) with terrain = geom - diff, z = avg_pool(terrain, k), mu = avg_pool(friction, k).
    `z` carries the interleaved (z, mu) pair for `DPhysics` (see `staged_pair`)."""
    _lib.require_hip_tensor(geom, 'geom')
    if geom.dtype != torch.float32:
        raise TypeError('stage_terrain: float32 only')
    shape = geom.shape
    sq = (lambda t: t.reshape(t.shape[0], t.shape[-2], t.shape[-1]).contiguous())  # noqa: E731
    terrain, z, mu, zmu = _Stage.apply(sq(geom), sq(diff), sq(friction.to(torch.float32)), int(k))
    z._mf_staged = (zmu, mu)       # found by DPhysics when exactly these two tensors come in as (z_grid, friction)
    return terrain.view(shape), z, mu


def staged_pair(z_grid, friction):
    """The interleaved [B,h,w,2] pair `stage_terrain` left for (z_grid, friction), or None if they are not that pair any more
    (other tensors, or modified in place since)."""
    st = getattr(z_grid, '_mf_staged', None)
    if st is None or friction is None or st[1] is not friction or z_grid._version != 0 or friction._version != 0:
        return None
    return st[0]
