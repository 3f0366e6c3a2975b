"""Pooling helpers of the terrain encoder under the reference's names (`monoforce.models.terrain_encoder.utils`:
`gen_dx_bx` :136-141, `cumsum_trick` :144-152, `QuickCumsum` :155-181).

The reference pools rank-sorted rows with a float32 prefix sum and a first difference at the run ends; its accuracy decays
with the row index.  Here a run is summed on its own: `segment_ids` numbers the runs of equal rank and
the rows are added into their run's slot (`index_add_`), so every voxel sum is exact to one rounding per addend.  Same
call signatures and return values (sums in run order, `geom_feats` of each run's last row); the HIP splat
(`monoforce_amd.splat`) does not go through these at all -- they exist for callers of the reference API.
"""
import torch

__all__ = ['gen_dx_bx', 'cumsum_trick', 'QuickCumsum', 'segment_ids']


def gen_dx_bx(xbound, ybound, zbound):
    """Voxel size dx, first voxel centre bx and voxel counts nx of the BEV grid from (lo, hi, step) bounds.
    float32 / int64 like the reference; the count truncates the float quotient (`LongTensor` semantics)."""
    lo, hi, step = (torch.tensor([float(b[k]) for b in (xbound, ybound, zbound)], dtype=torch.float64) for k in range(3))
    nx = torch.tensor([int((b[1] - b[0]) / b[2]) for b in (xbound, ybound, zbound)], dtype=torch.int64)
    return step.float(), (lo + step / 2.0).float(), nx


def segment_ids(ranks):
    """(run index of every row, index of each run's last row, number of runs) for rank-sorted rows."""
    n = ranks.shape[0]
    end = torch.ones(n, dtype=torch.bool, device=ranks.device)           # a row closes its run if the next rank differs
    if n > 1:
        end[:-1] = ranks[1:] != ranks[:-1]
    last = torch.nonzero(end).flatten()
    seg = torch.cumsum(end, 0) - end.long()                              # runs closed before this row
    return seg, last, int(last.numel())


def _pool_runs(x, ranks):
    seg, last, n_seg = segment_ids(ranks)
    out = torch.zeros((n_seg,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    out.index_add_(0, seg, x)
    return out, seg, last


def cumsum_trick(x, geom_feats, ranks):
    """Per-run sums of rank-sorted rows and the `geom_feats` row that closes each run (reference API, utils.py:144-152);
    differentiable through `index_add_`."""
    out, _, last = _pool_runs(x, ranks)
    return out, geom_feats[last]


class QuickCumsum(torch.autograd.Function):
    """Same pooling with the explicit backward `grad_x[i] = grad_out[run(i)]` (reference API, utils.py:155-181)."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks):
        out, seg, last = _pool_runs(x, ranks)
        kept = geom_feats[last]
        ctx.save_for_backward(seg)
        ctx.mark_non_differentiable(kept)
        return out, kept

    @staticmethod
    def backward(ctx, grad_out, _grad_geom):
        seg, = ctx.saved_tensors
        return grad_out.index_select(0, seg), None, None
