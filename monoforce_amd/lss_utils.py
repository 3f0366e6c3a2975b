"""Helpers of the terrain encoder (mirror of `/root/reference/monoforce/src/monoforce/models/terrain_encoder/utils.py`,
pooling part :136-181; the image-augmentation helpers of that file are host-side preprocessing and out of scope).
"""
import torch

__all__ = ['gen_dx_bx', 'cumsum_trick', 'QuickCumsum']


def gen_dx_bx(xbound, ybound, zbound):
    """Voxel size dx, first voxel centre bx and voxel counts nx of the BEV grid (utils.py:136-141).  float32 / int64."""
    rows = (xbound, ybound, zbound)
    dx = torch.Tensor([r[2] for r in rows])
    bx = torch.Tensor([r[0] + r[2] / 2.0 for r in rows])
    nx = torch.LongTensor([(r[1] - r[0]) / r[2] for r in rows])
    return dx, bx, nx


def cumsum_trick(x, geom_feats, ranks):
    """Segmented sum of rank-sorted rows via prefix sum + first difference at run ends (utils.py:144-152).
    Kept for API compatibility (plain torch, any device); the HIP splat does NOT use it -- it sums each voxel directly,
    which is ~1000x more accurate in float32."""
    x = x.cumsum(0)
    last = torch.ones(x.shape[0], device=x.device, dtype=torch.bool)
    last[:-1] = ranks[1:] != ranks[:-1]
    x, geom_feats = x[last], geom_feats[last]
    return torch.cat((x[:1], x[1:] - x[:-1])), geom_feats


class QuickCumsum(torch.autograd.Function):
    """`cumsum_trick` with the hand-written backward `grad_x[i] = grad_out[run(i)]` (utils.py:155-181)."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks):
        x = x.cumsum(0)
        last = torch.ones(x.shape[0], device=x.device, dtype=torch.bool)
        last[:-1] = ranks[1:] != ranks[:-1]
        x, geom_feats = x[last], geom_feats[last]
        x = torch.cat((x[:1], x[1:] - x[:-1]))
        ctx.save_for_backward(last)
        ctx.mark_non_differentiable(geom_feats)
        return x, geom_feats

    @staticmethod
    def backward(ctx, gradx, gradgeom):
        last, = ctx.saved_tensors
        run = torch.cumsum(last, 0)
        run[last] -= 1
        return gradx[run], None, None
