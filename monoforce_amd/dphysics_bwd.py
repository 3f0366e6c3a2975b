"""Backward of the fused rollout: marshals `_RolloutFn`'s saved tensors into `mf_rollout_bwd_*` (include/monoforce_hip.h).

Replaces the T x ~300-node autograd graph of the reference (`loss.backward()` through
`/root/reference/monoforce/src/monoforce/models/traj_predictor/dphysics.py:172-272,467-528`) with one kernel launch.
"""
import ctypes as C

import torch

from . import _lib, _timing

import os

# private copies of a shared map's gradient (2 x 256 KiB each at 256x256); MF_GRAD_COPIES overrides (tuning)
# (c3, 1024 rollouts x 4 points, streaming backward: 16 / 32 / 64 copies -> kernel 0.218 / 0.210 / 0.207 ms, step 0.3865 / 0.3831 /
#  0.3887 ms -- the reduction over the copies grows with them: 32)
GRAD_COPIES = int(os.environ.get('MF_GRAD_COPIES', '32'))
# LDS-window launches (0: as everywhere else).  Fit step at 16 384 / 32 768 rollouts, backward kernel at 256 / 64 / 32 copies: 0.969 / 1.004 /
# 1.059 and 1.524 / 1.536 / 1.632 ms (the windows' final adds meet in fewer copies) against 0.125 / 0.035 / 0.02 ms of reduction: 64
# (profiles/r6c_ab_step_copies.txt)
WIN_GRAD_COPIES = int(os.environ.get('MF_GRAD_COPIES_WIN', '64'))


def grad_copies_for(B, N):
    """Private gradient copies of a shared map for B rollouts of an N-point body.  Bodies of up to 4 points: ~64 rollouts per copy
    between GRAD_COPIES (32) and 256.  Larger bodies put N points of every
    rollout on nearly the same cells (the rollouts of a batch start from one pose): at least 64 copies -- measured with
    tools/ab_grad_copies.py, backward kernel at 16 / 64 / 256 copies: 256 x 223 points 1.20 / 0.94 / 0.88 ms, 1024 x 32 points
    0.98 / 0.81 / 0.78 ms, 64 x 223 0.91 / 0.85 / 0.85 ms (one copy per rollout) -- 64, not 256: the zero fill and the
    reduction over the copies grow with them (2 x 256 KiB each at 256 x 256)."""
    # (... and more with the number of points in flight: 2048 x 32 points 1.33 ms at 64 copies, 1.19 at 256; 1024 x 64: 0.86 / 0.77)
    floor = GRAD_COPIES if N <= 4 else max(GRAD_COPIES, min(256, max(64, (B * N) // 512)))
    return max(1, min(max(floor, B // 64), 256, B))


class GradPool:
    """The private gradient copies of a shared-map backward, [n_maps][copies][H*W] + 16 zeros (the row absent upstream
    gradients point at), kept ZEROED between steps: `mf_reduce_grad_copies_*` sums the copies and clears them in one launch, so a
    step pays neither the zero fill nor a separate reduction (~10 us of dependent-launch latency each at the BASELINE shape).
    `busy` is set while a backward is between its kernel launch and the reduction; a pool found busy (an exception in between,
    or two streams at once) is not trusted and gets refilled.
    `pinned`: a hipGraph capture used this pool -- its address is baked into the graph, so the pool must outlive every replay
    (`grad_pool` never evicts it).  `done` is an event recorded behind the last reduction: a later acquire on ANOTHER stream
    waits for it before it clears the buffer."""

    def __init__(self, n_maps, copies, n, dt, dev):
        self.n_maps, self.copies, self.n = n_maps, copies, n
        self.buf = torch.zeros(n_maps * copies * n + 16, dtype=dt, device=dev)
        self.busy = False
        self.stream = None
        self.pinned = False
        self.done = None

    def acquire(self):
        cur_s = torch.cuda.current_stream(self.buf.device)
        cur = cur_s.cuda_stream
        capturing = torch.cuda.is_current_stream_capturing()
        self.pinned = self.pinned or capturing
        if self.busy or (self.stream is not None and self.stream != cur):
            if self.done is not None and not capturing:
                cur_s.wait_event(self.done)       # the other stream's scatter / reduction may still be in flight
            self.buf.zero_()
        self.busy, self.stream = True, cur

    def reduce(self, map_shape):
        from .dphysics import _scalar_suffix
        # one spare scalar behind the maps: a data-parallel caller puts its loss there and exchanges gradients and loss with
        # ONE collective on this buffer, no pack / unpack (train.py::TerrainFitProblem)
        flat = torch.empty(self.n_maps * self.n + 1, dtype=self.buf.dtype, device=self.buf.device)
        out = flat[:-1].view((self.n_maps,) + tuple(map_shape))
        fn = getattr(_lib.lib(), 'mf_reduce_grad_copies_' + _scalar_suffix(self.buf.dtype))
        with torch.cuda.device(self.buf.device):
            _lib.check(fn(_lib.ptr(self.buf), self.n_maps, self.copies, C.c_longlong(self.n), _lib.ptr(out),
                          C.c_void_p(self.stream)), 'mf_reduce_grad_copies')
        if not torch.cuda.is_current_stream_capturing():
            if self.done is None:
                self.done = torch.cuda.Event()
            self.done.record(torch.cuda.current_stream(self.buf.device))
        self.busy = False
        return out


MAX_IDLE_POOLS = 4      # per owner; pools a captured graph references are kept on top of these


def grad_pool(owner, n_maps, copies, n, dt, dev):
    """The owner's (a DPhysics module's) pool for this shape, acquired for one backward.  Shapes come and go (tests, sweeps), so
    the least recently used pools beyond MAX_IDLE_POOLS are dropped -- never one a hipGraph capture has used (`pinned`: the
    graph replays into its address) and never one between its scatter and its reduction."""
    pools = owner.__dict__.setdefault('_grad_pools', {})
    key = (n_maps, copies, n, dt, dev)
    p = pools.pop(key, None)
    if p is None:
        idle = [k for k, q in pools.items() if not q.pinned and not q.busy]
        for k in idle[:max(len(idle) - (MAX_IDLE_POOLS - 1), 0)]:       # dicts keep insertion order: oldest first
            del pools[k]
        p = GradPool(n_maps, copies, n, dt, dev)
    pools[key] = p                                 # (re-)inserted last = most recently used
    p.acquire()
    return p

def rollout_backward(ctx, gXs, gXds, gRs, gOm, gFs, gFf, gloss=None):
    # (policy queries -- mf_rollout_bwd_wants_gcontrols -- read the CU count of the CURRENT device: the tensors' device, throughout)
    with torch.cuda.device(ctx.saved_tensors[0].device):
        return _rollout_backward_on_device(ctx, gXs, gXds, gRs, gOm, gFs, gFf, gloss)


def _rollout_backward_on_device(ctx, gXs, gXds, gRs, gOm, gFs, gFf, gloss):
    from .dphysics import _scalar_suffix, _stream_ptr
    controls, x_init, xd0, R0, w0, ts, Xraw, Xds, Rs, Om = ctx.saved_tensors
    desc, keep, mod = ctx.desc, ctx.keep, ctx.mod
    dev, dt = controls.device, controls.dtype
    B, T = desc.B, desc.T
    tm = desc.layout == _lib.MF_LAYOUT_TIME_MAJOR

    def up(g):     # upstream gradient -> the kernel's layout, contiguous; None stays NULL (= zeros)
        if g is None:
            return None
        g = g.to(dt)
        return (g.transpose(0, 1) if tm else g).contiguous()

    ups = [up(g) for g in (gXs, gXds, gRs, gOm, gFs, gFf)]
    z, mu = keep['z'], keep['mu']
    want_gmu = mu is not None and ctx.needs_input_grad[2]
    if desc.map_shared:
        # private gradient copies: rollout b scatters into copy b % copies, summed below (same-address atomics serialise)
        # ~64 rollouts per copy (same-address atomics serialise), between GRAD_COPIES and 256 copies
        copies = grad_copies_for(B, desc.N)
        # ... unless the launch sends its cell gradients through per-workgroup LDS windows (mf_rollout_bwd_window: saturated positions-only
        # launches): a workgroup adds its window to ONE copy once, at its end -- 256 copies cost a 0.125 ms reduction for 0.035 ms of kernel time
        if (WIN_GRAD_COPIES and dt == torch.float32 and (gloss is not None or ups[0] is not None) and all(u is None for u in ups[1:])
                and _lib.lib().mf_rollout_bwd_window(C.byref(desc))):
            copies = min(copies, WIN_GRAD_COPIES)
        desc.grad_copies = copies
        # one zero fill for [gz copies | gmu copies | the zero row absent upstream gradients point at]
        n_maps = 2 if want_gmu else 1
        pool = grad_pool(mod, n_maps, copies, z.numel(), dt, dev)
        maps = pool.buf[:n_maps * copies * z.numel()].view((n_maps, copies) + tuple(z.shape))
        gz, gmu, zero_row = maps[0], (maps[1] if want_gmu else None), pool.buf[-16:]
    else:
        maps = None
        gz = torch.zeros_like(z)
        gmu = torch.zeros_like(mu) if want_gmu else None
        zero_row = torch.zeros(16, dtype=dt, device=dev)
    # the control gradient is skipped where nobody wants it and the chosen kernels can leave it out (mf_rollout_bwd_wants_gcontrols)
    need_gc = ctx.needs_input_grad[3] or bool(_lib.lib().mf_rollout_bwd_wants_gcontrols(C.byref(desc)))      # (round 6: never forced by the library)
    gcontrols = torch.empty_like(controls) if need_gc else None
    gxd0, gR0, gw0 = torch.empty_like(xd0), torch.empty_like(R0), torch.empty_like(w0)
    gx0 = torch.empty_like(xd0) if ctx.needs_input_grad[4] else None
    ja = getattr(ctx, 'joint_angles', None)
    # (zeros: the default integrator never reads its last row of angles, and the kernel does not write that gradient row)
    gja = torch.zeros_like(ja) if ja is not None and ctx.needs_input_grad[10] else None
    bufs = _lib.MfRolloutBwdBufs(
        z=_lib.ptr(z), mu=_lib.ptr(mu), controls=_lib.ptr(controls), ts=_lib.ptr(ts), points=_lib.ptr(keep['points']),
        part=_lib.ptr(mod._part_dev(dev)), x_init=_lib.ptr(x_init), xd0=_lib.ptr(xd0), R0=_lib.ptr(R0), w0=_lib.ptr(w0),
        Xraw=_lib.ptr(Xraw), Xds=_lib.ptr(Xds), Rs=_lib.ptr(Rs), Omegas=_lib.ptr(Om),
        gXs=_lib.ptr(ups[0]), gXds=_lib.ptr(ups[1]), gRs=_lib.ptr(ups[2]), gOmegas=_lib.ptr(ups[3]),
        gFs=_lib.ptr(ups[4]), gFf=_lib.ptr(ups[5]), zeros=_lib.ptr(zero_row),
        gz=_lib.ptr(gz), gmu=_lib.ptr(gmu), gcontrols=_lib.ptr(gcontrols), gx0=_lib.ptr(gx0),
        gxd0=_lib.ptr(gxd0), gR0=_lib.ptr(gR0), gw0=_lib.ptr(gw0), joint_angles=_lib.ptr(ja), gjoint_angles=_lib.ptr(gja),
        rec=_lib.ptr(getattr(ctx, 'rec', None)), zmu_scratch=_lib.ptr(getattr(ctx, 'zmu', (None, None))[0]), zmu=_lib.ptr(getattr(ctx, 'zmu', (None, None))[1]))
    if gloss is not None:       # the forward carried physics_loss itself (MfRolloutLoss): the kernel forms dL/dXs from Xs and the ground truth
        spec, X_gt, Xs_rows, loss_out = ctx.loss
        gl = gloss.to(dt).reshape(1).contiguous()
        lstruct = _lib.MfRolloutLoss(T2=spec.T2, gt=_lib.ptr(X_gt), near=_lib.ptr(spec.near), w=_lib.ptr(spec.w), row_stamp=_lib.ptr(spec.row_stamp), row_w=_lib.ptr(spec.row_w),
                                     gloss=_lib.ptr(gl), Xs=_lib.ptr(Xs_rows))
        if loss_out is not None:       # MF_LOSS_VALUE_IN_BACKWARD: this launch also forms the value the forward left as NaN
            lstruct.flags = _lib.MF_LOSS_VALUE_IN_BACKWARD
            lstruct.partial, lstruct.ticket, lstruct.loss = _lib.ptr(loss_out[1]), _lib.ptr(spec.ticket(dev, torch.cuda.current_stream(dev))), _lib.ptr(loss_out[0])
        bufs.loss = C.cast(C.pointer(lstruct), C.c_void_p)
    fn = getattr(_lib.lib(), 'mf_rollout_bwd_' + _scalar_suffix(dt))
    with torch.cuda.device(dev), _timing.timed('rollout_bwd_kernel', dev):
        _lib.check(fn(C.byref(desc), C.byref(bufs), _stream_ptr(dev)), 'mf_rollout_bwd')
    _timing.note_launch('rollout_bwd_kernel')

    if desc.map_shared:
        summed = pool.reduce(z.shape)          # one launch: both maps summed over their copies, the pool left zeroed
        gz, gmu = summed[0], (summed[1] if want_gmu else None)

    def to_input_shape(g, shape, expanded):
        """Gradient of a map input in that input's own shape.  A per-rollout map given where the kernels ran shared maps
        cannot happen (`_make_desc` expands); what can: a shared run ([H,W] gradient g) whose input was [1,H,W] -> g[None],
        or ONE map expanded over the batch (stride 0) -> autograd's ExpandBackward sums whatever [B,H,W] gradient it is
        handed, so it gets g/B as a stride-0 expand when B is a power of two (the division and the B-fold sum are then
        exact), and otherwise g in row 0 and zeros elsewhere (exact for every B, costs the B x H x W buffer).  A per-rollout
        run whose input was one shared map (the other map was per-rollout) gets the sum over the rollouts."""
        if g is None:
            return None
        Bm = shape[0]
        if g.dim() == 3:                         # per-rollout gradient [B,H,W]
            if Bm == g.shape[0] and not expanded:
                return g
            gs_ = g.sum(0)                       # the input was ONE map ([1,H,W] or an expand of it)
            if Bm == 1:
                return gs_.unsqueeze(0)
            g = gs_
        if Bm == 1:
            return g.unsqueeze(0)
        if Bm & (Bm - 1) == 0:
            return (g / Bm).unsqueeze(0).expand(shape)
        full = torch.zeros(shape, dtype=g.dtype, device=g.device)
        full[0] = g
        return full

    return (None, to_input_shape(gz, ctx.z_shape, ctx.z_expanded) if ctx.needs_input_grad[1] else None,
            to_input_shape(gmu, ctx.mu_shape, ctx.mu_expanded), gcontrols if ctx.needs_input_grad[3] else None, gx0,
            gxd0 if ctx.needs_input_grad[5] else None, gR0 if ctx.needs_input_grad[6] else None,
            gw0 if ctx.needs_input_grad[7] else None, None, None, gja)
