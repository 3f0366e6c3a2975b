"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests).

Rollouts never interact (every op of the step is per batch row), so the forward shards the batch with
no communication at all.  The backward has exactly one exchange: the gradient of whatever is SHARED between ranks --
a shared terrain/friction grid (2 x 256 x 256 x 4 B = 512 KiB) or the encoder parameters -- summed with one flat,
bucketed all-reduce.  xGMI is point-to-point (7 links per GPU); payloads here are latency- not bandwidth-bound, so a
single flat bucket (one collective launch) is the right shape.
"""
import os

import torch
import torch.distributed as dist

__all__ = ['init', 'world', 'rank', 'shard_range', 'shard', 'allreduce_sum_', 'FlatBucket']


def init(backend=None, device=None):
    """Initialise the default process group from torchrun's env (RANK / WORLD_SIZE / MASTER_*); no-op for 1 process."""
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1 or dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    kw = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
    dist.init_process_group(backend, **kw)


def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_range(n, r=None, w=None):
    """Contiguous [lo, hi) slice of n units owned by rank r of w (first n % w ranks get one extra)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, extra = divmod(n, w)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard(t, dim=0):
    """This rank's rows of a batch-major tensor (None passes through)."""
    if t is None or world() == 1:
        return t
    lo, hi = shard_range(t.shape[dim])
    return t.narrow(dim, lo, hi - lo)


class FlatBucket:
    """Persistent flat buffer over a list of tensors so that their sum across ranks is ONE collective."""

    def __init__(self, tensors):
        self.shapes = [t.shape for t in tensors]
        self.numels = [t.numel() for t in tensors]
        self.buf = torch.empty(sum(self.numels), dtype=tensors[0].dtype, device=tensors[0].device)

    def pack(self, tensors):
        o = 0
        for t, n in zip(tensors, self.numels):
            self.buf[o:o + n].copy_(t.reshape(-1))
            o += n
        return self.buf

    def unpack_into(self, tensors):
        o = 0
        for t, n in zip(tensors, self.numels):
            t.copy_(self.buf[o:o + n].view(t.shape))
            o += n


def allreduce_sum_(tensors, bucket=None, average=False):
    """In-place sum (or mean) of `tensors` across ranks through one flat bucket; returns the bucket for reuse."""
    tensors = [t for t in tensors if t is not None]
    if world() == 1 or not tensors:
        return bucket
    if bucket is None:
        bucket = FlatBucket(tensors)
    buf = bucket.pack(tensors)
    if buf.is_cuda and dist.get_backend() == 'gloo':      # CPU-only test rigs: stage through the host
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)        # RCCL over xGMI
    if average:
        buf.div_(world())
    bucket.unpack_into(tensors)
    return bucket
