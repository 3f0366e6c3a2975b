"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests).

Rollouts never interact (every op of the step is per batch row), so the forward shards the batch with
no communication at all.  The backward has exactly one exchange: the gradient of whatever is SHARED between ranks.
  * a shared terrain/friction grid (2 x 256 x 256 x 4 B = 512 KiB): one flat bucket, one all-reduce (`allreduce_sum_`) --
    latency-bound on xGMI, nothing to overlap with;
  * the encoder parameters (~55 MB): `GradBuckets` -- two 32 MB buckets whose all-reduces start from autograd hooks while
    the rest of the backward is still running.  xGMI is point-to-point (7 links per GPU, ring collectives are per-link
    bound), so few large collectives, not many small ones.
"""
import os

import torch
import torch.distributed as dist

__all__ = ['init', 'world', 'rank', 'active', 'shard_range', 'shard', 'allreduce_sum_', 'FlatBucket', 'GradBuckets']

# MF_DIST_FORCE=1: run every collective even in a process group of ONE rank.  A one-GPU box can then execute the exact RCCL
# code path of the multi-GPU step (init_process_group('nccl', device_id=...), all_reduce(AVG) in place, the hooked bucket
# exchange, a hipGraph replay followed by a collective on the same stream) -- tests/test_dist_nccl_gpu.py, bench.py
# MF_BENCH_FORCE_DIST=1 -- instead of the first 8-GPU run being the first run of that code.
FORCE = os.environ.get('MF_DIST_FORCE', '0') not in ('', '0')


def init(backend=None, device=None, force=None):
    """Initialise the default process group from torchrun's env (RANK / WORLD_SIZE / MASTER_*); no-op for 1 process unless
    `force` (a one-rank group; default: FORCE, i.e. MF_DIST_FORCE -- a forced group also makes `active()` true)."""
    global FORCE
    force = FORCE if force is None else bool(force)
    FORCE = FORCE or force
    if (int(os.environ.get('WORLD_SIZE', '1')) <= 1 and not force) or dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    kw = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
    dist.init_process_group(backend, **kw)


def active():
    """True when collectives have to run: more than one rank, or a one-rank group under MF_DIST_FORCE / `FORCE`."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


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


def shared_flat_buffer(tensors):
    """A 1-D view over the storage `tensors` are consecutive slices of, when that storage holds exactly them plus ONE spare
    element behind the last -- or None.  (The shared-map rollout backward hands its map gradients out that way,
    dphysics_bwd.GradPool.reduce; autograd detaches them on the way into `.grad`, so this goes by storage, not by `_base`.)"""
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return None
    st, isz, off = tensors[0].untyped_storage(), tensors[0].element_size(), 0
    for t in tensors:
        if (t.untyped_storage().data_ptr() != st.data_ptr() or not t.is_contiguous() or t.dtype != tensors[0].dtype
                or t.storage_offset() != off):
            return None
        off += t.numel()
    if st.nbytes() != (off + 1) * isz:
        return None
    return torch.as_strided(tensors[0], (off + 1,), (1,), 0)


def allreduce_mean_inplace_(flat):
    """Mean of one buffer across ranks, in place, as ONE collective (RCCL computes the average itself: no divide launch)."""
    if not active():
        return flat
    if flat.is_cuda and dist.get_backend() == 'gloo':      # CPU-only test rigs: stage through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host / world())
    elif dist.get_backend() == 'gloo':
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world())
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)        # RCCL over xGMI
    return flat


def allreduce_sum_(tensors, bucket=None, average=False):
    """In-place sum (or mean) of `tensors` across ranks through one flat bucket; returns the bucket for reuse."""
    tensors = [t for t in tensors if t is not None]
    if not active() or not tensors:
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


def _allreduce_sum_async(buf):
    """Start the sum of one flat buffer across ranks; returns the work handle (None where the exchange already completed)."""
    if buf.is_cuda and dist.get_backend() == 'gloo':      # CPU-only test rigs: stage through the host, synchronously
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        buf.copy_(host)
        return None
    return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)        # RCCL over xGMI, on its own stream


class GradBuckets:
    """Data-parallel gradient exchange of a parameter list, overlapped with the backward pass (encoder weights, BASELINE
    config 5: ~14 M parameters = 55 MB per step over RCCL).

    The parameters are assigned, in reverse registration order (roughly the order their gradients become ready), to a few
    flat buffers of about `bucket_mb` megabytes.  A post-accumulate hook counts the ready gradients of a bucket; when the
    last one lands they are packed into the buffer with ONE multi-tensor copy and the bucket's all-reduce is launched
    asynchronously (RCCL runs it on its own stream, concurrently with the rest of the backward).  `finish()` waits for the
    collectives, averages, and points every `p.grad` at its slice of the reduced buffer (no copy back).  xGMI is point-to-point
    and each collective pays its launch and ring latency, so few, large buckets (default 32 MB -> two for the encoder)
    rather than many small ones.

    `zero()` replaces `optimizer.zero_grad()`: it drops the gradients (`None`), so autograd hands each new gradient over
    without an accumulation kernel.  Parameters that receive no gradient in a step are exchanged as zeros and keep
    `grad = None`.
    """

    def __init__(self, params, bucket_mb=32.0, average=True):
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.defer = False       # True: hooks and `pack()` only pack; the collectives run in `exchange()` (graph-replayed steps)
        self.buckets = []
        cap = int(bucket_mb * (1 << 20))
        group, nbytes = [], 0
        for p in reversed(self.params):
            group.append(p)
            nbytes += p.numel() * p.element_size()
            if nbytes >= cap:
                self._make_bucket(group)
                group, nbytes = [], 0
        if group:
            self._make_bucket(group)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _make_bucket(self, group):
        p0 = group[0]
        buf = torch.zeros(sum(p.numel() for p in group), dtype=p0.dtype, device=p0.device)
        views, o = [], 0
        for p in group:
            views.append(buf[o:o + p.numel()].view_as(p))
            p._mf_bucket = len(self.buckets)
            o += p.numel()
        self.buckets.append(dict(buf=buf, params=list(group), views=views, ready=0, work=None, launched=False))

    def zero(self):
        """Start of a step: forget the gradients and re-arm the ready counters."""
        for b in self.buckets:
            b['ready'], b['work'], b['launched'] = 0, None, False
            for p in b['params']:
                p.grad = None

    def _launch(self, b):
        b['launched'] = True
        have = [(v, p.grad) for v, p in zip(b['views'], b['params']) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        # parameters the forward did not use (the same ones on every rank: same model, same code path) are exchanged as zeros
        # and keep `grad = None` afterwards, like in the reference, so the optimizer skips them (Adam's weight decay would
        # otherwise move them on a zero gradient)
        b['got'] = [p.grad is not None for p in b['params']]
        for v, p in zip(b['views'], b['params']):
            if p.grad is None:
                v.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])      # one multi-tensor pack
        if active() and not self.defer:
            b['work'] = _allreduce_sum_async(b['buf'])

    def _on_grad(self, p):
        b = self.buckets[p._mf_bucket]
        b['ready'] += 1
        if b['ready'] == len(b['params']) and not b['launched'] and not self.defer:
            self._launch(b)

    def pack(self):
        """After `loss.backward()`: pack (and, unless deferred, start exchanging) every bucket the hooks have not launched."""
        for b in self.buckets:
            if not b['launched']:
                self._launch(b)

    def exchange(self):
        """All-reduce every bucket as it stands and wait: the exchange of a deferred step (a step replayed as hipGraphs packs
        its buckets inside the first graph; collectives are launched live between the graphs), and `comm_ms` of bench.py."""
        if not active():
            return
        works = [_allreduce_sum_async(b['buf']) for b in self.buckets]
        for w in works:
            if w is not None:
                w.wait()

    def finish(self):
        """After `loss.backward()`: launch what has not been launched, wait, average; `p.grad` then views the reduced buffers."""
        self.pack()
        for b in self.buckets:
            if b['work'] is not None:
                b['work'].wait()
                b['work'] = None
            if self.average and world() > 1:
                b['buf'].div_(world())
            for v, p, got in zip(b['views'], b['params'], b['got']):
                p.grad = v if got else None

    def remove(self):
        for h in self._hooks:
            h.remove()
