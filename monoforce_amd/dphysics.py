"""DPhysics: differentiable-physics trajectory rollout on the MI355X HIP kernels.

Host-side mirror of `/root/reference/monoforce/src/monoforce/models/traj_predictor/dphysics.py` (class `DPhysics`
:144-605 and the module-level helpers).  Same constructor, same `forward(z_grid, controls, joint_angles=None,
state=None, vis=False, friction=None)` signature, same return structure
`((Xs, Xds, Rs, Omegas), (F_springs, F_frictions))`, same shape asserts and messages, same side effects
(`state[0][:, 2]` is overwritten with the terrain height, `self.ts` is truncated to the controls' length).
What differs is where the work happens: the whole T-step scan is ONE kernel launch through the C ABI
(`mf_rollout_fwd_*`, include/monoforce_hip.h); there is no CPU path.

Outputs are `[B, T, ...]` *views* of time-major buffers -- exactly the layout the reference's default integrator
returns (`dphysics.py:515-526` permutes torchdiffeq's `[T, B, ...]` stacks); pass `contiguous_outputs=True` to the
constructor to get batch-major contiguous tensors instead.
"""
import ctypes as C
import os

import torch

from . import _lib, _timing
from .dphys_config import DPhysConfig

__all__ = ['DPhysics', 'generate_controls', 'vw_to_track_vels', 'inertia_tensor', 'normalized', 'skew_symmetric']


# ------------------------------------------------------------------------------------------------------------
# module-level helpers of the reference API (plain torch; tiny, host side)
# ------------------------------------------------------------------------------------------------------------
def normalized(x, eps=1e-6, dim=-1):
    """x / max(|x|, eps) (dphysics.py:7-19)."""
    return x / torch.clamp(torch.norm(x, dim=dim, keepdim=True), min=eps)


def skew_symmetric(v):
    """[v]x for v[B,3] (dphysics.py:22-40)."""
    assert v.dim() == 2 and v.shape[1] == 3
    U = torch.zeros(v.shape[0], 3, 3, device=v.device, dtype=v.dtype)
    U[:, 0, 1], U[:, 0, 2], U[:, 1, 2] = -v[:, 2], v[:, 1], -v[:, 0]
    U[:, 1, 0], U[:, 2, 0], U[:, 2, 1] = v[:, 2], -v[:, 1], v[:, 0]
    return U


def generate_controls(n_trajs=10, time_horizon=5.0, dt=0.01, v_range=(-1.0, 1.0), w_range=(-1.0, 1.0)):
    """Constant-in-time (v, w) per trajectory, uniformly sampled; returns ([n, N, 2], stamps[N]) (dphysics.py:42-72)."""
    N = int(time_horizon / dt)
    stamps = torch.linspace(0, time_horizon, N)
    v = torch.rand(n_trajs) * (v_range[1] - v_range[0]) + v_range[0]
    w = torch.rand(n_trajs) * (w_range[1] - w_range[0]) + w_range[0]
    vw = torch.stack([v, w], dim=-1)                       # [n, 2]
    return vw.unsqueeze(1).repeat(1, N, 1), stamps


def vw_to_track_vels(v, w, robot_size, n_tracks):
    """(v, w) -> track speeds: [L, R] or [FL, FR, RL, RR] (dphysics.py:75-104)."""
    Ly = robot_size[1]
    if n_tracks == 2:
        return torch.stack([v - w * (Ly / 2.0), v + w * (Ly / 2.0)], dim=-1)
    if n_tracks == 4:
        lo, hi = v - w * Ly / 2.0, v + w * Ly / 2.0
        return torch.stack([lo, hi, lo, hi], dim=-1)
    raise ValueError('n_tracks must be 2 or 4')


def inertia_tensor(mass, points):
    """Inertia [B,3,3] of N equal point masses (total `mass`) about the body origin, points[B,N,3] (dphysics.py:107-141)."""
    assert points.dim() == 3
    mp = mass / points.shape[1]
    x, y, z = points.unbind(-1)
    d = [(mp * (y ** 2 + z ** 2)).sum(1), (mp * (x ** 2 + z ** 2)).sum(1), (mp * (x ** 2 + y ** 2)).sum(1)]
    xy, xz, yz = -(mp * x * y).sum(1), -(mp * x * z).sum(1), -(mp * y * z).sum(1)
    I = torch.stack([torch.stack([d[0], xy, xz], 1), torch.stack([xy, d[1], yz], 1), torch.stack([xz, yz, d[2]], 1)], 1)
    assert I.shape == (points.shape[0], 3, 3)
    return I


# ------------------------------------------------------------------------------------------------------------
# the op
# ------------------------------------------------------------------------------------------------------------
def _scalar_suffix(dtype):
    if dtype == torch.float32:
        return 'f32'
    if dtype == torch.float64:
        return 'f64'
    raise TypeError(f'DPhysics computes in float32 or float64, got {dtype}')


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _is_shared_map(t):
    """True if a [B,H,W] map is one [H,W] map broadcast over the batch (`expand`, stride 0) -- read once, not B times."""
    return t.dim() == 3 and (t.shape[0] == 1 or t.stride(0) == 0)


def _default_state(controls, B, in_kernel=False):
    """x = 0, xd = (v_0, 0, 0), R = I, omega = (0, 0, w_0) (dphysics.py:554-559).  xd and omega stay differentiable functions
    of the first control when it requires grad (as in the reference); otherwise one launch (mf_rollout_default_state_*), or
    none: with `in_kernel` the buffers come back UNINITIALISED and the rollout kernel fills them (MfRolloutDesc.default_state).
    Returns (x, xd, R, w, filled_by_rollout_kernel)."""
    if not (controls.requires_grad and torch.is_grad_enabled()) and controls.is_cuda and (
            controls.is_contiguous() or (in_kernel and _time_constant(controls))):
        buf = torch.empty(18 * B, dtype=controls.dtype, device=controls.device)
        x, xd, R, w = buf[:3 * B].view(B, 3), buf[3 * B:6 * B].view(B, 3), buf[6 * B:15 * B].view(B, 3, 3), buf[15 * B:].view(B, 3)
        if in_kernel:
            return x, xd, R, w, True
        fn = getattr(_lib.lib(), 'mf_rollout_default_state_' + _scalar_suffix(controls.dtype))
        with torch.cuda.device(controls.device):
            _lib.check(fn(C.c_int32(B), C.c_int32(controls.shape[1]), _lib.ptr(controls), _lib.ptr(x), _lib.ptr(xd), _lib.ptr(R), _lib.ptr(w),
                          _stream_ptr(controls.device)), 'mf_rollout_default_state')
        return x, xd, R, w, False
    buf = torch.zeros(12 * B, dtype=controls.dtype, device=controls.device)
    x, R = buf[:3 * B].view(B, 3), buf[3 * B:].view(B, 3, 3)
    R.view(B, 9)[:, ::4] = 1.0
    first = controls[:, 0]
    xd = torch.nn.functional.pad(first[:, 0:1], (0, 2))
    w = torch.nn.functional.pad(first[:, 1:2], (2, 0))
    return x, xd, R, w, False


def _time_constant(controls):
    """True for a [B,T,2] view of ONE (v, w) per rollout broadcast over time (`expand`, stride 0 along T): the kernels read it
    as it is (MfRolloutDesc.controls_stride_t = 0) -- no [B,T,2] copy for every step of every rollout to fetch."""
    return (controls.dim() == 3 and controls.shape[1] > 1 and controls.stride(1) == 0 and controls.stride(2) == 1
            and controls.stride(0) >= 2)


def _kernel_controls(controls, allow_view):
    """(tensor whose memory the kernel reads, stride over rollouts, stride over time); (t, 0, 0) = contiguous."""
    if allow_view and controls.is_cuda and _time_constant(controls):
        return controls, controls.stride(0), 0
    return controls.contiguous(), 0, 0


def _zmu_scratch(mod, desc, z):
    """Scratch for the interleaved (z, mu) copy of a SHARED float32 map pair (MfRolloutFwdBufs.zmu_scratch): the library uses
    it where its kernels gain from it and ignores it elsewhere; stream-ordered, so it may be freed right after the launch."""
    if mod.interleave_maps and desc.map_shared and (z.dtype == torch.float32 or mod.points_per_lane == _lib.MF_LANES_COMPONENT):
        return torch.empty(2 * desc.H * desc.W, dtype=z.dtype, device=z.device)
    return None


class LossSpec:
    """Ground-truth stamps of `physics_loss` (losses.py:102-127) prepared for the kernels that carry the loss themselves
    (MfRolloutLoss): the SAME stamp times for every rollout.  near[j] = output row nearest in time to stamp j (losses.py:116),
    w[j] = 1 / (1 + gamma t_j), row_stamp = the inverse table.  Built once per time grid (one small host round trip); the tables
    live in persistent device buffers that `refresh_` rebuilds in place from a new batch's stamps WITHOUT a host round trip, so a
    captured train step can carry the rebuild (a replay then reads the stamps the caller copied into the batch tensors)."""

    def __init__(self, pred_ts, gt_ts, gamma, device, dtype=torch.float32):
        pred_ts = torch.as_tensor(pred_ts, dtype=dtype).reshape(-1).cpu()
        gt_ts = torch.as_tensor(gt_ts, dtype=dtype).reshape(-1).cpu()
        near = (pred_ts.unsqueeze(0) - gt_ts.unsqueeze(1)).abs().argmin(dim=1)
        self.fusable = bool((near[1:] > near[:-1]).all()) if near.numel() > 1 else near.numel() == 1      # one stamp per row at most
        self.T, self.T2, self.gamma = int(pred_ts.numel()), int(gt_ts.numel()), float(gamma)
        row_stamp = torch.full((self.T,), -1, dtype=torch.int32)
        row_stamp[near] = torch.arange(self.T2, dtype=torch.int32)
        w = 1. / (1. + gamma * gt_ts)
        row_w = torch.zeros(self.T, dtype=dtype)
        row_w[near] = w
        self.near = near.to(torch.int32).to(device)
        self.w = w.to(device)
        self.row_stamp, self.row_w = row_stamp.to(device), row_w.to(device)
        self.gt_ts = gt_ts.to(device)
        self.pred_ts = pred_ts.to(device)
        self._stamp_ids = torch.arange(self.T2, dtype=torch.int32, device=device)
        # 0, or NaN once `refresh_` met stamps the fused kernels cannot carry (two stamps on one output row, rows that differ between
        # the rollouts): callers add it to the loss, so such a batch fails LOUDLY instead of being scored against stale tables
        self.poison = torch.zeros((), dtype=dtype, device=device)
        self._tickets = {}
        self._full = {}

    def ticket(self, device, stream):
        """The zero-initialised launch counter of the loss reductions, one per (device, stream) like `losses._ticket`: launches
        ordered on one stream share it (the kernel leaves it zero); two streams using one spec (a capture stream beside the default
        one) no longer tick the same counter."""
        from . import losses
        losses._register_reset()
        t = losses._ticket(device, stream)
        self._tickets[(device.index, stream.cuda_stream)] = t      # a captured graph replays into this address: it lives as long as the spec
        return t

    def refresh_(self, gt_ts):
        """Rebuild the tables in place from new stamps `gt_ts` ([T2], or [B,T2] with every row the same): device ops only, no host
        synchronisation, capturable.  Returns self."""
        rows = gt_ts if gt_ts.dim() == 2 else gt_ts.unsqueeze(0)
        row = rows[0].to(device=self.gt_ts.device, dtype=self.gt_ts.dtype)
        assert row.numel() == self.T2, f'{row.numel()} stamps, this LossSpec was built for {self.T2}'
        near = (self.pred_ts.unsqueeze(0) - row.unsqueeze(1)).abs().argmin(dim=1)
        self.near.copy_(near)
        self.gt_ts.copy_(row)
        torch.reciprocal(1. + self.gamma * row, out=self.w)
        self.row_stamp.fill_(-1).scatter_(0, near, self._stamp_ids)
        self.row_w.zero_().scatter_(0, near, self.w)
        bad = (near[1:] <= near[:-1]).any() if self.T2 > 1 else torch.zeros((), dtype=torch.bool, device=row.device)
        if rows.shape[0] > 1 and rows.stride(0) != 0:
            bad = bad | (rows != rows[:1]).any()
        self.poison.copy_(torch.where(bad, float('nan'), 0.))
        for B, (near_b, ts_b) in self._full.items():
            near_b.copy_(self.near.unsqueeze(0).expand(B, -1))
            ts_b.copy_(self.gt_ts.unsqueeze(0).expand(B, -1))
        return self

    def per_rollout(self, B):
        """(nearest [B,T2] int32, gt_ts [B,T2]) as `mf_physics_loss_value_*` reads them (one row per rollout), built once per batch size."""
        if B not in self._full:
            self._full[B] = (self.near.unsqueeze(0).expand(B, -1).contiguous(), self.gt_ts.unsqueeze(0).expand(B, -1).contiguous())
        return self._full[B]


def _rollout_forward(ctx, mod, z, mu, controls, x_arg, xd0, R0, w0, ts, want_grad, joint_angles=None, want_forces=True,
                     x0_buf=None, x0_private=False, default_state=False, loss=None):
    """One `mf_rollout_fwd_*` launch; `loss` = (LossSpec, X_gt[B,T2,3]) fuses physics_loss into it (outs then start with the loss)."""
    # (the library's policy queries -- record bytes, force stride, who stages the interleaved maps -- read the CU count of the CURRENT
    #  device: everything runs with the tensors' device current, not only the launch itself.  ADVICE r4.)
    with torch.cuda.device(z.device):
        return _rollout_forward_on_device(ctx, mod, z, mu, controls, x_arg, xd0, R0, w0, ts, want_grad, joint_angles, want_forces,
                                          x0_buf, x0_private, default_state, loss)


def _rollout_forward_on_device(ctx, mod, z, mu, controls, x_arg, xd0, R0, w0, ts, want_grad, joint_angles, want_forces,
                               x0_buf, x0_private, default_state, loss):
    # x_arg is the autograd input (the caller's start position when it requires grad); the kernel works on x0_buf, the
    # detached contiguous buffer that receives the snapped height.  x0_private: nobody else sees that buffer.
    x0 = x0_buf if x0_buf is not None else x_arg
    # inference reads a time-constant (expanded) control tensor as it is; the backward kernel wants [B][T][2]
    controls, sb, st = _kernel_controls(controls, allow_view=not want_grad)
    desc, keep = mod._make_desc(z, mu, controls)
    desc.controls_stride_b, desc.controls_stride_t = sb, st
    desc.default_state = int(default_state)     # the kernel computes the start state and fills x0 / xd0 / R0 / w0
    if joint_angles is not None:
        desc.has_joints = 1
        for i, v in enumerate(sum((list(p) for p in mod.dphys_cfg.joint_positions.values()), [])[:12]):
            desc.joint_xyz[i] = float(v)
    B, T, N = desc.B, desc.T, desc.N
    dt, dev = z.dtype, z.device
    tm = desc.layout == _lib.MF_LAYOUT_TIME_MAJOR
    lead = (T, B) if tm else (B, T)
    new = lambda *tail: torch.empty(*lead, *tail, dtype=dt, device=dev)  # noqa: E731
    Np = _lib.lib().mf_rollout_force_stride(C.byref(desc))      # point slots per force row the chosen kernel writes
    if Np < N:
        raise RuntimeError('mf_rollout_force_stride rejected the descriptor')
    desc.force_stride = Np
    Xs, Xds, Rs, Om = new(3), new(3), new(3, 3), new(3)
    Fs, Ff = (new(Np, 3), new(Np, 3)) if want_forces else (None, None)
    Xraw = new(3) if want_grad else None
    # the per-step record of the component-parallel kernels (MfRolloutFwdBufs.rec): kept for the backward where the library
    # says it pays (few rollouts of a small body), 1 KiB per rollout and step
    rec = None
    if want_grad and joint_angles is None:      # (float64: the validation build of the component-parallel kernels, 32-byte quads)
        rb = _lib.lib().mf_rollout_record_bytes if dt == torch.float32 else _lib.lib().mf_rollout_record_bytes_f64
        nbytes = int(rb(C.byref(desc)))
        if nbytes > 0:
            rec = torch.empty(nbytes // dt.itemsize, dtype=dt, device=dev)
    zmu_scratch, zmu_staged = _zmu_scratch(mod, desc, keep['z']), mod._staged_zmu(desc, z, mu)
    bufs = _lib.MfRolloutFwdBufs(
        z=_lib.ptr(keep['z']), mu=_lib.ptr(keep['mu']), controls=_lib.ptr(controls), ts=_lib.ptr(ts),
        points=_lib.ptr(keep['points']), part=_lib.ptr(mod._part_dev(dev)),
        x0=_lib.ptr(x0), xd0=_lib.ptr(xd0), R0=_lib.ptr(R0), w0=_lib.ptr(w0),
        Xs=_lib.ptr(Xs), Xds=_lib.ptr(Xds), Rs=_lib.ptr(Rs), Omegas=_lib.ptr(Om), Fs=_lib.ptr(Fs), Ff=_lib.ptr(Ff),
        Xraw=_lib.ptr(Xraw), joint_angles=_lib.ptr(joint_angles),
        zmu_scratch=_lib.ptr(zmu_scratch), zmu=_lib.ptr(zmu_staged), rec=_lib.ptr(rec))
    loss_val = lstruct = None
    in_backward = loss is not None and len(loss) > 2 and bool(loss[2]) and want_grad      # MF_LOSS_VALUE_IN_BACKWARD
    saturated = loss is not None and len(loss) > 3 and loss[3] == 2      # mf_rollout_loss_fusable = 2: the fusion lives in the BACKWARD launch alone
    in_forward = (loss is not None and mod.loss_in_forward and not in_backward and not saturated and not want_forces
                  and desc.integrator == _lib.MF_INTEG_ODEINT_EULER)      # (the LOSS kernels: default integrator, component-parallel)
    if loss is not None:
        spec, X_gt = loss[:2]
        loss_val = torch.empty((), dtype=dt, device=dev)
    if in_backward and saturated:      # (the saturated forward kernels carry no loss at all: the value is marked as not yet known here)
        loss_val.fill_(float('nan'))
    elif in_backward:     # the launch only marks the value as not yet known (NaN); mf_rollout_bwd_* fills it
        lstruct = _lib.MfRolloutLoss(T2=spec.T2, flags=_lib.MF_LOSS_VALUE_IN_BACKWARD, loss=_lib.ptr(loss_val))
        bufs.loss = C.cast(C.pointer(lstruct), C.c_void_p)
    if in_forward:
        partial = torch.empty((B + 3) // 4, dtype=dt, device=dev)
        lstruct = _lib.MfRolloutLoss(T2=spec.T2, gt=_lib.ptr(X_gt), near=_lib.ptr(spec.near), w=_lib.ptr(spec.w), row_stamp=_lib.ptr(spec.row_stamp), row_w=_lib.ptr(spec.row_w),
                                     partial=_lib.ptr(partial), ticket=_lib.ptr(spec.ticket(dev, torch.cuda.current_stream(dev))), loss=_lib.ptr(loss_val))
        bufs.loss = C.cast(C.pointer(lstruct), C.c_void_p)       # (lstruct stays alive until the launch call below returns)
    fn = getattr(_lib.lib(), 'mf_rollout_fwd_' + _scalar_suffix(dt))
    with torch.cuda.device(dev), _timing.timed('rollout_fwd_kernel', dev):
        _lib.check(fn(C.byref(desc), C.byref(bufs), _stream_ptr(dev)), 'mf_rollout_fwd')
    _timing.note_launch('rollout_fwd_kernel')
    outs = (Xs, Xds, Rs, Om) + ((Fs[..., :N, :], Ff[..., :N, :]) if want_forces else ())
    if tm:
        outs = tuple(o.transpose(0, 1) for o in outs)
    if loss is not None and not in_forward and not in_backward:
        # The VALUE of the loss from one small launch on the rows just written (mf_physics_loss_value_*: gather-reduce over the B x T2
        # stamped rows, the mean finished inside it).  Measured against the rollout kernel accumulating it itself (LOSS kernels,
        # `loss_in_forward`): that forward pays ~15 instructions and two loads at EVERY step of a launch bound by the issue slots of
        # one wave per SIMD -- 0.165 -> 0.185 ms at B = 1024 -- where this launch is 0.012 ms.  The BACKWARD half of the fusion is
        # the one that pays: no loss-gradient launch, no [T][B][3] gradient rows (rollout_backward).
        near_b, gt_ts_b = spec.per_rollout(B)
        Xp = outs[0]
        ldesc = _lib.MfLossDesc(B=B, T1=T, T2=spec.T2, x_stride_b=Xp.stride(0), x_stride_t=Xp.stride(1), gamma=spec.gamma)
        lpart = torch.empty((B * spec.T2 + 255) // 256, dtype=dt, device=dev)
        with torch.cuda.device(dev), _timing.timed('physics_loss_fwd', dev):
            _lib.check(getattr(_lib.lib(), 'mf_physics_loss_value_' + _scalar_suffix(dt))(
                C.byref(ldesc), _lib.ptr(Xp), _lib.ptr(X_gt), _lib.ptr(gt_ts_b), _lib.ptr(near_b), _lib.ptr(lpart), _lib.ptr(spec.ticket(dev, torch.cuda.current_stream(dev))),
                _lib.ptr(loss_val), None, C.c_longlong(0), _stream_ptr(dev)), 'mf_physics_loss_value')
    if loss is not None:
        outs = (loss_val,) + outs
        # (the backward launch's scratch is allocated HERE: under a hipGraph capture the autograd thread must not allocate)
        # (... and the scalar is held through an ALIAS: the tensor handed out gets this node as its grad_fn, and a node holding its own
        #  output is a reference cycle -- garbage that the collector then frees whenever it runs, e.g. in the middle of a later capture)
        # (partial sums of the direction that forms the value: one per workgroup -- B / 4 component-parallel, at most B one point per lane)
        ctx.loss = (spec, X_gt, Xs, (loss_val.detach(), torch.empty(B if saturated else (B + 3) // 4, dtype=dt, device=dev)) if in_backward else None) if want_grad else None
    ctx.n_force_outs = 2 if want_forces else 0
    # outputs the loss does not touch arrive as None in backward (= NULL upstream pointers), not as zero-filled tensors
    ctx.set_materialize_grads(False)
    if want_grad:
        ctx.mod, ctx.desc, ctx.keep = mod, desc, keep
        # (the backward's record-reading kernel reads the shared maps interleaved too: the staged pair, or this scratch, which it fills itself)
        ctx.zmu = (zmu_scratch, zmu_staged) if keep['mu'] is not None else (None, None)
        if ctx.zmu[1] is None and zmu_scratch is not None and dt == torch.float32 and _lib.lib().mf_rollout_fwd_stages_zmu(C.byref(desc)):
            ctx.zmu = (None, zmu_scratch)       # this launch filled the scratch: the backward reads it as it is
        ctx.z_shape, ctx.mu_shape, ctx.mu_given = z.shape, (mu.shape if mu is not None else None), mu is not None
        ctx.z_expanded = z.stride(0) == 0 and z.shape[0] > 1
        ctx.mu_expanded = mu is not None and mu.stride(0) == 0 and mu.shape[0] > 1
        ctx.joint_angles = joint_angles.detach() if joint_angles is not None else None
        ctx.rec = rec
        # x0 now holds the snapped start position; a caller-visible buffer is copied, the module's own default is not
        ctx.save_for_backward(controls, x0 if x0_private else x0.clone(), xd0, R0, w0, ts, Xraw, Xds, Rs, Om)
    return outs


class _RolloutFn(torch.autograd.Function):
    """forward = mf_rollout_fwd_*; backward = mf_rollout_bwd_* (reverse-time adjoint of the same scan)."""

    @staticmethod
    def forward(ctx, mod, z, mu, controls, x_arg, xd0, R0, w0, ts, want_grad, joint_angles=None, want_forces=True,
                x0_buf=None, x0_private=False, default_state=False):
        return _rollout_forward(ctx, mod, z, mu, controls, x_arg, xd0, R0, w0, ts, want_grad, joint_angles, want_forces,
                                x0_buf, x0_private, default_state)

    @staticmethod
    def backward(ctx, gXs, gXds, gRs, gOm, gFs=None, gFf=None):
        from .dphysics_bwd import rollout_backward
        return rollout_backward(ctx, gXs, gXds, gRs, gOm, gFs, gFf) + (None, None, None, None)


class _RolloutLossFn(torch.autograd.Function):
    """The rollout with `physics_loss` inside both launches (MfRolloutLoss): outputs (loss, Xs, Xds, Rs, Omegas), of which only the
    loss is differentiable -- the backward forms dL/dXs itself at the stamped rows."""

    @staticmethod
    def forward(ctx, mod, z, mu, controls, x_arg, xd0, R0, w0, ts, want_grad, x0_buf, x0_private, default_state, spec, X_gt,
                value_in_backward=False, route=1, want_forces=False):
        outs = _rollout_forward(ctx, mod, z, mu, controls, x_arg, xd0, R0, w0, ts, want_grad, None, bool(want_forces), x0_buf, x0_private,
                                default_state, loss=(spec, X_gt, value_in_backward, route))
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, gloss, *_unused):
        from .dphysics_bwd import rollout_backward
        if gloss is None:
            return (None,) * 18
        grads = rollout_backward(ctx, None, None, None, None, None, None, gloss=gloss)      # (mod, z, mu, controls, x, xd0, R0, w0, ts, want_grad, ja)
        return grads[:10] + (None, None, None, None, None, None, None, None)


class DPhysics(torch.nn.Module):
    """Drop-in for the reference `DPhysics` (dphysics.py:144); no parameters, all state is configuration."""

    def __init__(self, dphys_cfg=None, device='cpu', contiguous_outputs=False, block=0, snap_to_terrain=True,
                 points_per_lane=0, precise=False, return_forces=True):
        super().__init__()
        self.dphys_cfg = dphys_cfg if dphys_cfg is not None else DPhysConfig()
        self.device = device
        self.x_points = self.dphys_cfg.robot_points.to(self.device).unsqueeze(0)         # (1, N, 3)
        self.I = inertia_tensor(mass=self.dphys_cfg.robot_mass, points=self.x_points)
        self.I_inv = torch.linalg.inv(self.I)
        self.z_grid = None
        self.friction = None
        self.stiffness = self.dphys_cfg.stiffness
        self.damping = self.dphys_cfg.damping
        self.controls = None
        self.joint_angles = None
        # time grid snapshot at construction, like the reference (dphysics.py:166-167)
        T, dt = self.dphys_cfg.traj_sim_time, self.dphys_cfg.dt
        self._ts_full_len = int(T / dt)
        self._ts_T = T
        self.ts = torch.linspace(0, T, self._ts_full_len).to(self.device)
        self.contiguous_outputs = contiguous_outputs
        self.block = block
        self.snap_to_terrain = snap_to_terrain     # False: continue from `state` as is (no reference equivalent)
        self.points_per_lane = points_per_lane     # kernel lane mapping: 0 auto, 1 latency-oriented, 4 throughput-oriented
        # False: skip the force outputs (forward returns (states, (None, None))): training only consumes the states
        # (scripts/train.py:243 `states_pred, _ = self.dphysics(...)`); 57 % less output traffic.  float32 fast math only.
        self.return_forces = return_forces
        self.interleave_maps = True      # shared float32 maps: let the library read an interleaved (z, mu) copy (same bits, fewer loads)
        self.precise = precise      # True: float32 kernels in the reference's exact op order (IEEE div/sqrt, no FMA); ~1.5x slower
        # physics_loss_rollout: True = the forward rollout kernel accumulates the loss itself (LOSS kernels); False (default) = its value
        # comes from one small launch on the written rows -- measured faster, see _rollout_forward; the backward is fused either way
        self.loss_in_forward = bool(int(os.environ.get('MF_LOSS_IN_FORWARD', '0')))
        self.staged_handoffs = 0    # rollouts that read an interleaved map pair staged by terrain_stage.stage_terrain
        self._cache = {}

    # -- constants marshalled for the C ABI ---------------------------------------------------------------
    def _part_dev(self, dev):
        key = ('part', str(dev))
        if key not in self._cache:
            N = self.x_points.shape[1]
            part = torch.full((N,), -1, dtype=torch.int32)
            for j, mask in enumerate(self.dphys_cfg.driving_parts):       # later masks overwrite earlier (:243-246)
                m = torch.as_tensor(mask).cpu()
                assert m.dim() == 1 and m.shape[0] == N
                part[m] = j
            self._cache[key] = part.to(dev)
        return self._cache[key]

    def _points_dev(self, dev, dtype):
        key = ('pts', str(dev), dtype)
        if key not in self._cache:
            self._cache[key] = self.dphys_cfg.robot_points.to(device=dev, dtype=dtype).contiguous()
        return self._cache[key]

    def _iinv(self, dtype):
        key = ('iinv', dtype)
        if key not in self._cache:
            P = self.dphys_cfg.robot_points.detach().cpu().to(dtype).unsqueeze(0)
            self._cache[key] = torch.linalg.inv(inertia_tensor(self.dphys_cfg.robot_mass, P))[0].double().flatten().tolist()
        return self._cache[key]

    def _staged_zmu(self, desc, z, mu):
        """The interleaved (z, mu) pair `terrain_stage.stage_terrain` produced together with exactly these two maps (ONE shared
        map pair, float32), for MfRolloutFwdBufs.zmu; None otherwise."""
        if not (self.interleave_maps and desc.map_shared and z.dtype == torch.float32 and z.shape[0] == 1):
            return None
        from .terrain_stage import staged_pair
        zmu = staged_pair(z, mu)
        if zmu is None or zmu.shape[0] != 1:
            return None
        self.staged_handoffs += 1
        return zmu

    def _make_desc(self, z, mu, controls):
        cfg = self.dphys_cfg
        B = controls.shape[0]
        assert z.dim() == 3, f'z_grid must be [B,H,W] (or [1,H,W] = one map shared by all rollouts), got {tuple(z.shape)}'
        _, H, W = z.shape
        for name, m in (('z_grid', z), ('friction', mu)):
            if m is None:
                continue
            assert m.dim() == 3 and tuple(m.shape[-2:]) == (H, W), \
                f'{name} shape {tuple(m.shape)} does not match the {H}x{W} height grid'
            assert m.shape[0] in (1, B), f'{name} batch {m.shape[0]} is neither 1 (shared map) nor the {B} rollouts of controls'
        shared = _is_shared_map(z) and (mu is None or _is_shared_map(mu))
        if shared:
            zc = z[0].contiguous()
            muc = None if mu is None else mu[0].contiguous()
        else:
            # one map shared, the other per rollout: the kernels index both at b*H*W, so the shared one is expanded for real
            zc = z.expand(B, H, W).contiguous()
            muc = None if mu is None else mu.expand(B, H, W).contiguous()
        integ = _lib.MF_INTEG_ODEINT_EULER if cfg.use_odeint else _lib.MF_INTEG_DYNAMICS
        if cfg.integration_mode != 'euler':
            # the reference's 'rk4' is a degenerate formula on the custom loop and an adaptive-free torchdiffeq solver on
            # the other; neither is part of the parity scope
            raise ValueError(f'Unknown integration mode: {cfg.integration_mode}' if cfg.integration_mode != 'rk4' else
                             "integration_mode 'rk4' is not supported by the HIP rollout (only 'euler')")
        desc = _lib.MfRolloutDesc(
            B=controls.shape[0], T=controls.shape[1], N=self.x_points.shape[1], H=H, W=W,
            n_tracks=len(cfg.driving_parts), integrator=integ,
            layout=_lib.MF_LAYOUT_BATCH_MAJOR if self.contiguous_outputs else _lib.MF_LAYOUT_TIME_MAJOR,
            map_shared=int(shared), block=self.block, skip_snap=int(not self.snap_to_terrain), points_per_lane=self.points_per_lane,
            math_mode=_lib.MF_MATH_EXACT if self.precise else _lib.MF_MATH_FAST,
            mass=float(cfg.robot_mass), gravity=float(cfg.gravity), stiffness=float(self.stiffness),
            damping=float(self.damping), omega_max=float(cfg.omega_max), grid_res=float(cfg.grid_res),
            d_max=float(cfg.d_max), dt=float(cfg.dt), robot_size_y=float(cfg.robot_size[1]))
        for i, v in enumerate(self._iinv(z.dtype)):
            desc.Iinv[i] = v
        keep = dict(z=zc, mu=muc, points=self._points_dev(z.device, z.dtype))
        return desc, keep

    # -- reference API --------------------------------------------------------------------------------------
    def dphysics(self, z_grid, controls, joint_angles=None, state=None, friction=None, _loss=None):
        """Simulate the robot on the terrain (dphysics.py:530-594).

        z_grid (B,H,W), controls (B,N,2), joint_angles (B,N,4) or None, state=(x,xd,R,omega) or None, friction (B,H,W).
        Returns ((Xs,Xds,Rs,Omegas), (F_springs,F_frictions)).
        """
        cfg = self.dphys_cfg
        dev = torch.device(self.device)
        dt_, T_ = cfg.dt, cfg.traj_sim_time
        # the drop-in step at the fused step's speed (api_cache.py): after a few identical forward -> physics_loss cycles the whole step is
        # ONE hipGraph replay; every other call runs launch by launch below (and is observed)
        api_key = None
        if _loss is None:
            cache = self.__dict__.get('_api_step_cache')
            if cache is None:
                from .api_cache import ApiStepCache
                cache = self.__dict__['_api_step_cache'] = ApiStepCache(self)
            api_key = cache.forward_key(z_grid, controls, friction, joint_angles, state)
            if api_key is not None:
                hit = cache.try_forward(api_key, z_grid, controls, friction)
                if hit is not None:
                    return hit
        # extension over the reference: a [1,H,W] map with B > 1 controls is ONE terrain shared by all rollouts
        batch_size = z_grid.shape[0] if z_grid.shape[0] != 1 else controls.shape[0]
        z_grid = z_grid.to(dev)
        _lib.require_hip_tensor(z_grid, 'z_grid')
        dtype = z_grid.dtype
        _scalar_suffix(dtype)        # float32 / float64 only: anything else is refused here, loudly
        controls = controls.to(device=dev, dtype=dtype)

        own_state = state is None
        state_in_kernel = False
        if own_state:                                                                # (:554-559)
            if not _time_constant(controls):
                controls = controls.contiguous()
            *state, state_in_kernel = _default_state(controls, batch_size, in_kernel=True)
        if friction is not None:
            friction = friction.to(device=dev, dtype=dtype)
        # The attributes the reference keeps (:561-580) hold DETACHED tensors here: a view of a leaf that requires grad, kept
        # alive on the module, keeps that leaf's gradient accumulator -- and the stream it was created on -- alive too, and a
        # later hipGraph capture of a backward to the same leaf on another stream then dies in hipStreamEndCapture
        # (tools/try_graph_step.py; round 1's "core dump when capturing a train step").
        self.z_grid = z_grid.detach()
        self.friction = friction.detach() if friction is not None else cfg.friction      # all-ones default (:562), never expanded

        N_ts = min(int(T_ / dt_), controls.shape[1])                                 # (:573)
        B = state[0].shape[0]
        assert controls.shape == (B, N_ts, 2), f'Controls shape {controls.shape} != {(B, N_ts, 2)}'
        self.controls = controls.detach()
        ja_dev = None
        if joint_angles is not None:
            assert joint_angles.shape == (B, N_ts, 4), f'Joint angles shape {joint_angles.shape} != {(B, N_ts, 4)}'
            # the reference re-articulates the body only for robot == 'marv' and non-zero angles (dphysics.py:340)
            if cfg.robot == 'marv' and not torch.allclose(joint_angles, torch.zeros_like(joint_angles)):
                # (differentiable: the gradient flows through update_joints and the per-step inertia, dphysics.py:191-197)
                ja_dev = joint_angles.to(device=dev, dtype=dtype).contiguous()
        self.joint_angles = joint_angles.detach() if joint_angles is not None else None
        self.ts = self.ts[:N_ts]                                                     # permanent, like the reference (:581)
        ts = self._time_grid(N_ts, dtype, dev)

        # kernel inputs: contiguous, right dtype; x0's z component is written by the kernel (:567-571)
        x_in = state[0]
        x0 = x_in.detach().to(device=dev, dtype=dtype).contiguous()
        aliased = x0.data_ptr() == x_in.data_ptr()
        if aliased and x_in.requires_grad:
            x0 = x0.clone(); aliased = False
        xd0, R0, w0 = (s.to(device=dev, dtype=dtype).contiguous() for s in state[1:])
        want_grad = torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (z_grid, friction, controls, x_in, xd0, R0, w0, ja_dev))
        # the float64 validation build of the fast kernels: a states-only forward exists where the component-parallel kernels run
        # (N <= 4) or a multi-wave record is kept for a backward (mf_rollout_record_bytes_f64 > 0) -- elsewhere float64 needs the force
        # buffers (mf_rollout_fwd_f64), so they stay (N > 4 under no_grad, or B beyond the record range)
        cp64 = dtype == torch.float64 and self.points_per_lane == _lib.MF_LANES_COMPONENT and ja_dev is None
        if cp64 and self.x_points.shape[1] > 4:
            dq = _lib.MfRolloutDesc(B=B, T=N_ts, N=self.x_points.shape[1], H=z_grid.shape[-2], W=z_grid.shape[-1], n_tracks=len(cfg.driving_parts),
                                    integrator=_lib.MF_INTEG_ODEINT_EULER if cfg.use_odeint else _lib.MF_INTEG_DYNAMICS,
                                    points_per_lane=self.points_per_lane, math_mode=_lib.MF_MATH_FAST)
            with torch.cuda.device(dev):
                cp64 = want_grad and int(_lib.lib().mf_rollout_record_bytes_f64(C.byref(dq))) > 0
        want_forces = self.return_forces or self.precise or (dtype != torch.float32 and not cp64) or ja_dev is not None
        # a start position that requires grad is the autograd input itself (its gradient: x and y through the contact geometry,
        # z none -- the snap overwrites it); the kernel works on the detached buffer x0 either way
        x_arg = x_in if (want_grad and x_in.requires_grad) else x0
        loss_val = None
        if _loss is not None:       # physics_loss inside the launches (physics_loss_rollout)
            spec, X_gt = _loss[:2]
            outs = _RolloutLossFn.apply(self, z_grid, friction, controls, x_arg, xd0, R0, w0, ts, want_grad, x0, own_state,
                                        state_in_kernel, spec, X_gt, len(_loss) > 2 and bool(_loss[2]), _loss[3] if len(_loss) > 3 else 1,
                                        len(_loss) > 4 and bool(_loss[4]))
            loss_val, outs = outs[0], outs[1:]
        else:
            outs = _RolloutFn.apply(self, z_grid, friction, controls, x_arg, xd0, R0, w0, ts, want_grad, ja_dev, want_forces,
                                    x0, own_state, state_in_kernel)
        if not aliased:
            with torch.no_grad():       # the reference's in-place write (through .data: no version bump on a tensor autograd saved)
                x_in.data[..., 2] = x0[..., 2].to(device=x_in.device, dtype=x_in.dtype)
        Xs, Xds, Rs, Omegas = outs[:4]
        F_springs, F_frictions = outs[4:] if len(outs) == 6 else (None, None)
        if _loss is not None:
            if len(_loss) > 4 and _loss[4]:
                return loss_val, (Xs, Xds, Rs, Omegas), (F_springs, F_frictions)
            return loss_val, (Xs, Xds, Rs, Omegas)
        if api_key is not None:
            cache.tag_eager_outputs(Xs, api_key, None)
        return (Xs, Xds, Rs, Omegas), (F_springs, F_frictions)

    def loss_spec(self, gt_ts, gamma=0.9, n_steps=None, dtype=torch.float32):
        """Prepare the ground-truth stamps of `physics_loss` for `physics_loss_rollout`: gt_ts [T2] (the same stamp times for every
        rollout), weights 1 / (1 + gamma t) (losses.py:122); the predicted stamps are this module's time grid."""
        n = len(self.ts) if n_steps is None else int(n_steps)
        return LossSpec(self._time_grid(n, dtype, torch.device('cpu')), gt_ts, gamma, torch.device(self.device), dtype=dtype)

    def physics_loss_rollout(self, z_grid, controls, X_gt, spec, state=None, friction=None, value_in_backward=False, want_forces=False):
        """`physics_loss(self(z_grid, controls, ...), [X_gt], pred_ts, gt_ts, gamma)` (losses.py:102-127, the position term the training
        scripts use: scripts/train.py:399-406, scripts/fit_terrain.py:53-62) with the loss INSIDE the rollout's two launches (SURVEY.md
        8f rank 1): the forward kernel accumulates the time-weighted squared error at the stamped rows while it writes them, the
        backward forms dL/dXs there itself -- no loss launches, no [B,T,3] gradient tensor.  X_gt [B,T2,3]; `spec` = self.loss_spec(gt_ts).
        Returns (loss, (Xs, Xds, Rs, Omegas)); the states come back detached from the graph (only the loss is differentiable) and are
        READ-ONLY until `loss.backward()` has run: the backward launch re-reads the Xs rows to form dL/dXs (they are not copied, and an
        in-place edit would not be noticed by autograd's version check).
        `want_forces`: also write and return the force rows, (loss, states, (F_springs, F_frictions)) -- what `forward` hands out (the cached
        drop-in step, `_ApiStepCache`).
        `value_in_backward` (MF_LOSS_VALUE_IN_BACKWARD): for a caller that ALWAYS calls `loss.backward()` next and reads the value
        only afterwards (a fit loop): the backward launch forms the value too -- the returned scalar is NaN until then -- and the
        step loses its one remaining loss launch.
        Where the library cannot fuse (mf_rollout_loss_fusable: 1 = float32 fast math, a rigid body of <= 4 points, <= 2048 rollouts --
        dynamics(): <= 1024; 2 = the saturated positions-only backward, > 8192 rollouts of a <= 4-point body, whose forward takes the value
        from one small launch on its rows; 3 = the early-recompute component-parallel backward, 4097 .. 8192 rollouts, likewise; 0 in between
        -- 2049 .. 4096 rollouts, dynamics() 1025 .. 4096: the loss's own two launches are cheaper there; several stamps on one row: never)
        the same value and gradient come from the unfused route."""
        from .losses import physics_loss_fused
        cp64 = z_grid.dtype == torch.float64 and self.points_per_lane == _lib.MF_LANES_COMPONENT      # the validation build of the fast kernels
        ok = (spec.fusable and not self.precise and (z_grid.dtype == torch.float32 or cp64) and spec.w.dtype == z_grid.dtype
              and not self.contiguous_outputs)
        B = controls.shape[0]
        if ok:
            ok = spec.T == min(int(self.dphys_cfg.traj_sim_time / self.dphys_cfg.dt), controls.shape[1])
        if ok:
            d = _lib.MfRolloutDesc(B=B, T=spec.T, N=self.x_points.shape[1], H=z_grid.shape[-2], W=z_grid.shape[-1],
                                   integrator=_lib.MF_INTEG_ODEINT_EULER if self.dphys_cfg.use_odeint else _lib.MF_INTEG_DYNAMICS,
                                   math_mode=_lib.MF_MATH_FAST, force_stride=max(self.x_points.shape[1], 4), map_shared=1, layout=_lib.MF_LAYOUT_TIME_MAJOR,
                                   points_per_lane=self.points_per_lane)
            with torch.cuda.device(torch.device(self.device)):
                fus = int(_lib.lib().mf_rollout_loss_fusable(C.byref(d)))
            # (2: the saturated positions-only backward forms dL/dXs itself -- and the value, with `value_in_backward`; else the value comes
            #  from one small launch on the forward's rows)
            ok = fus == 1 or (fus in (2, 3) and z_grid.dtype == torch.float32)
            if fus == 3:       # (the one-wave component-parallel backward forms dL/dXs; the value: one small launch on the forward's rows)
                value_in_backward = False
        if not ok:
            # (the loss reads the positions only: the forward writes the states, not the 24 N bytes of force rows per rollout-step -- for the
            #  reference's 223-point body 36 of 241 MB per launch; `return_forces` is restored for the module's other callers)
            keep, self.return_forces = self.return_forces, bool(want_forces)
            try:
                states, forces = self.dphysics(z_grid, controls, state=state, friction=friction)
            finally:
                self.return_forces = keep
            B_, T2 = X_gt.shape[:2]
            gt_ts = spec.gt_ts.unsqueeze(0).expand(B_, -1)
            loss = physics_loss_fused(states, [X_gt], None, gt_ts, gamma=spec.gamma, nearest=spec.near.unsqueeze(0).expand(B_, -1))
            return (loss, states, forces) if want_forces else (loss, states)
        Xg = X_gt.detach().to(device=torch.device(self.device), dtype=z_grid.dtype).contiguous()
        assert Xg.shape == (B, spec.T2, 3), f'X_gt shape {tuple(Xg.shape)} != {(B, spec.T2, 3)}'
        return self.dphysics(z_grid, controls, state=state, friction=friction, _loss=(spec, Xg, bool(value_in_backward), fus, bool(want_forces)))

    @torch.no_grad()
    def rollout_costs(self, z_grid, controls, state=None, friction=None, pose_stride=None, project=True):
        """Trajectory-shooting forward: roll out and return only what the reference's planners consume.

        The kernel's path-cost mode writes, per output row, `(R[2,0], R[2,1], R[2,2], std_points |F_spring|)` -- the inputs of
        `norm(F_springs).std(points).std(time)` (monoforce_node.py:91) and of the roll / pitch cost
        (monoforce_ros/nodes/diff_physics.py:263-266) -- and keeps every `pose_stride`-th pose (`poses[::pose_step]`,
        monoforce_node.py:35,115; default 0.5 s like the node) plus the final one: 16 B per rollout-step instead of 180.
        float32 fast-math rigid-body rollouts only; arguments as `forward` (a [1,H,W] map is shared by all rollouts).
        `project` (default integrator only): the rows carry the third row of the nearest rotation to the drifting R, which is
        what scipy's `from_matrix(R).as_euler()` reads roll and pitch from; the force cost does not need it (`project=False`).

        Returns dict(cost_rows [B,T,4], Xs [B,Tp,3], Rs [B,Tp,3,3], pose_steps [Tp], force_cost [B]) -- views of time-major
        buffers; `force_cost` is the node's path cost itself (the std over time taken in the kernel's registers).
        """
        cfg = self.dphys_cfg
        dev = torch.device(self.device)
        if self.precise:
            raise ValueError('rollout_costs uses the float32 fast-math kernels: construct DPhysics(precise=False)')
        z_grid = z_grid.to(dev)
        _lib.require_hip_tensor(z_grid, 'z_grid')
        if z_grid.dtype != torch.float32:
            raise TypeError('rollout_costs: float32 only')
        controls = controls.to(device=dev, dtype=torch.float32)
        B = z_grid.shape[0] if z_grid.shape[0] != 1 else controls.shape[0]
        N_ts = min(int(cfg.traj_sim_time / cfg.dt), controls.shape[1])
        assert controls.shape == (B, N_ts, 2), f'Controls shape {controls.shape} != {(B, N_ts, 2)}'
        state_in_kernel = False
        controls, sb, st = _kernel_controls(controls.detach(), allow_view=True)
        if state is None:                                                            # (dphysics.py:554-559)
            x0, xd0, R0, w0, state_in_kernel = _default_state(controls, B, in_kernel=True)
        else:
            x0 = state[0].detach().to(device=dev, dtype=torch.float32).clone()       # the snap writes x0.z: keep the caller's
            xd0, R0, w0 = (t.detach().to(device=dev, dtype=torch.float32).contiguous() for t in state[1:])
        if friction is not None:
            friction = friction.to(device=dev, dtype=torch.float32)
        ps = int(pose_stride) if pose_stride else max(int(0.5 / cfg.dt), 1)
        ts = self._time_grid(N_ts, torch.float32, dev)
        desc, keep = self._make_desc(z_grid, friction, controls)
        desc.controls_stride_b, desc.controls_stride_t = sb, st
        desc.layout, desc.pose_stride, desc.cost_project = _lib.MF_LAYOUT_TIME_MAJOR, ps, int(bool(project))
        desc.default_state = int(state_in_kernel)
        Tp = 1 + (N_ts - 1 + ps - 1) // ps
        rows = torch.empty(N_ts, B, 4, device=dev)
        force_cost = torch.empty(B, device=dev)
        Xs, Rs = torch.empty(Tp, B, 3, device=dev), torch.empty(Tp, B, 3, 3, device=dev)
        bufs = _lib.MfRolloutFwdBufs(
            z=_lib.ptr(keep['z']), mu=_lib.ptr(keep['mu']), controls=_lib.ptr(controls), ts=_lib.ptr(ts),
            points=_lib.ptr(keep['points']), part=_lib.ptr(self._part_dev(dev)),
            x0=_lib.ptr(x0), xd0=_lib.ptr(xd0), R0=_lib.ptr(R0), w0=_lib.ptr(w0),
            Xs=_lib.ptr(Xs), Xds=None, Rs=_lib.ptr(Rs), Omegas=None, Fs=None, Ff=None, Xraw=None, joint_angles=None,
            cost_rows=_lib.ptr(rows), path_cost=_lib.ptr(force_cost), zmu_scratch=_lib.ptr(_zmu_scratch(self, desc, keep['z'])))
        with torch.cuda.device(dev), _timing.timed('rollout_fwd_kernel', dev):
            _lib.check(_lib.lib().mf_rollout_fwd_f32(C.byref(desc), C.byref(bufs), _stream_ptr(dev)), 'mf_rollout_fwd')
        key = ('pose_steps', Tp, ps, N_ts, str(dev))
        if key not in self._cache:       # constant per configuration: no two extra launches per planning cycle
            self._cache[key] = torch.clamp(torch.arange(Tp, device=dev) * ps, max=N_ts - 1)
        steps = self._cache[key]
        return dict(cost_rows=rows.transpose(0, 1), Xs=Xs.transpose(0, 1), Rs=Rs.transpose(0, 1), pose_steps=steps, force_cost=force_cost)

    def interpolate_grid(self, grid, x_query, y_query, return_normals=False, return_cells=False):
        """The reference's public sampling method (dphysics.py:385-455), bug for bug, as one HIP launch
        (`mf_interpolate_grid_*`, the rollout kernels' own device code): grid (B,H,W), x_query / y_query (B,N) ->
        values (B,N) [, unit normals (B,N,3)].  `return_cells=True` (no reference equivalent) appends the clamped flat indices
        (B,N,4) of the four cells (c, f, l, fl) and the fractions (B,N,2).  No autograd through this entry (the rollout has its
        own backward)."""
        cfg = self.dphys_cfg
        dev = torch.device(self.device)
        grid = torch.as_tensor(grid).to(dev)
        _lib.require_hip_tensor(grid, 'grid')
        dt = grid.dtype
        sfx = _scalar_suffix(dt)
        if any(torch.is_tensor(t) and t.requires_grad for t in (grid, x_query, y_query)) and torch.is_grad_enabled():
            raise RuntimeError('DPhysics.interpolate_grid: not differentiable on its own (gradients flow through forward())')
        B, H, W = grid.shape
        xq = torch.as_tensor(x_query).to(device=dev, dtype=dt).reshape(B, -1).contiguous()
        yq = torch.as_tensor(y_query).to(device=dev, dtype=dt).reshape(B, -1).contiguous()
        assert xq.shape == yq.shape, f'x_query {tuple(xq.shape)} and y_query {tuple(yq.shape)} differ'
        N = xq.shape[1]
        gc = grid.detach().contiguous()
        desc = _lib.MfInterpDesc(B=B, N=N, H=H, W=W, map_shared=0, math_mode=_lib.MF_MATH_EXACT if self.precise else _lib.MF_MATH_FAST,
                                 grid_res=float(cfg.grid_res), d_max=float(cfg.d_max))
        z = torch.empty(B, N, dtype=dt, device=dev)
        n = torch.empty(B, N, 3, dtype=dt, device=dev) if return_normals else None
        cells = torch.empty(B, N, 4, dtype=torch.int32, device=dev) if return_cells else None
        frac = torch.empty(B, N, 2, dtype=dt, device=dev) if return_cells else None
        with torch.cuda.device(dev):
            _lib.check(getattr(_lib.lib(), 'mf_interpolate_grid_' + sfx)(C.byref(desc), _lib.ptr(gc), _lib.ptr(xq), _lib.ptr(yq), _lib.ptr(z),
                                                                         _lib.ptr(n), _lib.ptr(cells), _lib.ptr(frac), _stream_ptr(dev)),
                       'mf_interpolate_grid')
        out = (z,) + ((n,) if return_normals else ()) + ((cells, frac) if return_cells else ())
        return out[0] if len(out) == 1 else out

    def _time_grid(self, n, dtype, dev):
        key = ('ts', n, dtype, str(dev))
        if key not in self._cache:
            self._cache[key] = torch.linspace(0, self._ts_T, self._ts_full_len, dtype=dtype)[:n].to(dev).contiguous()
        return self._cache[key]

    def forward(self, z_grid, controls, joint_angles=None, state=None, vis=False, friction=None):
        states, forces = self.dphysics(z_grid=z_grid, controls=controls, joint_angles=joint_angles, state=state,
                                       friction=friction)
        if vis:
            with torch.no_grad():
                self.visualize(states=states, z_grid=z_grid)
        return states, forces

    def visualize(self, states, z_grid, forces=None, states_gt=None, friction=None):
        """Matplotlib stand-in for the reference's mayavi animation (dphysics.py:607-669): plots one rollout's path."""
        import matplotlib.pyplot as plt
        Xs = states[0].detach().cpu().numpy()
        b = 0
        plt.figure()
        plt.imshow(z_grid[b].detach().cpu().numpy().T, origin='lower',
                   extent=[-self.dphys_cfg.d_max, self.dphys_cfg.d_max] * 2, cmap='terrain')
        plt.plot(Xs[b, :, 0], Xs[b, :, 1], 'k-')
        if states_gt is not None:
            G = states_gt[0].detach().cpu().numpy()
            plt.plot(G[b, :, 0], G[b, :, 1], 'b--')
        plt.xlabel('x [m]'); plt.ylabel('y [m]')
        plt.show()
