"""`torch.ops.monoforce.*`: the hot-path kernels registered with `torch.library` ("C-ABI the replacement exports").

Schemas (tensors on the GPU, float32 or float64; `Bz` in {1, B}: a [1,H,W] map is ONE map shared by all rollouts):

    monoforce::dphys_rollout_fwd(Tensor z, Tensor? mu, Tensor controls, Tensor x0, Tensor xd0, Tensor R0, Tensor w0,
                                 Tensor pts, Tensor part_id, Tensor Iinv, float[] consts, int integrator, bool save_for_bwd)
                                 -> (Tensor Xs, Tensor Xds, Tensor Rs, Tensor Om, Tensor Fs, Tensor Ff, Tensor Xraw, Tensor x0_snapped,
                                     Tensor rec)
    monoforce::dphys_rollout_bwd(Tensor z, Tensor? mu, Tensor controls, Tensor x_init, Tensor xd0, Tensor R0, Tensor w0,
                                 Tensor pts, Tensor part_id, Tensor Iinv, float[] consts, int integrator,
                                 Tensor Xraw, Tensor Xds, Tensor Rs, Tensor Om, Tensor rec,
                                 Tensor? gXs, Tensor? gXds, Tensor? gRs, Tensor? gOm, Tensor? gFs, Tensor? gFf)
                                 -> (Tensor gz, Tensor gmu, Tensor gcontrols, Tensor gx0, Tensor gxd0, Tensor gR0, Tensor gw0)
    monoforce::bev_splat_plan(Tensor geom, float[] dx, float[] bx, int[] nx) -> Tensor plan
    monoforce::bev_splat_fwd(Tensor x, Tensor plan, int B, int n_per_sample, float[] dx, float[] bx, int[] nx) -> Tensor
    monoforce::bev_splat_bwd(Tensor grad, Tensor plan, int B, int n_per_sample, int C, float[] dx, float[] bx, int[] nx) -> Tensor

`consts` = [mass, gravity, stiffness, damping, grid_res, d_max, dt, omega_max, robot_size_y(, traj_sim_time)]; `part_id[N]` int32
(index of the last driving mask holding the point, -1 = not driving); `Iinv` [3,3] on the HOST (nine scalars of the launch
descriptor); `integrator` 0 = dynamics(), 1 = odeint-euler (MF_INTEG_*).  The ops are functional: the start position with
its z component moved onto the terrain (dphysics.py:567-571, an in-place write in the reference) comes back as `x0_snapped`,
and `rollout()` below copies it into the caller's tensor.  Outputs are `[B,T,...]` views of time-major buffers, like
`DPhysics.forward`.  Every op is one or two launches of the C ABI (include/monoforce_hip.h) on
the current stream; no synchronisation, no host round trip, so they can be captured into a hipGraph.  Autograd formulas are
registered (`rollout_fwd` -> `rollout_bwd`, `bev_splat_fwd` -> `bev_splat_bwd`), as are shape functions for tracing.  `rec` is
the per-step record small launches keep for the backward (MfRolloutFwdBufs.rec; empty otherwise).

`DPhysics` itself keeps calling the C ABI through its own autograd function (monoforce_amd/dphysics.py) -- same library calls,
more options (articulated bodies, path costs, strided controls); `rollout()` / `splat()` below are the functional entries.
"""
import ctypes as C

import torch

from . import _lib, _timing
from .dphysics_bwd import grad_copies_for, grad_pool


class _PoolOwner:       # the registered ops have no module to hang the persistent gradient-copy pools on
    pass


_POOL_OWNER = _PoolOwner()

__all__ = ['rollout', 'splat', 'CONST_NAMES']

CONST_NAMES = ('mass', 'gravity', 'stiffness', 'damping', 'grid_res', 'd_max', 'dt', 'omega_max', 'robot_size_y')      # + optional traj_sim_time

_L = torch.library.Library('monoforce', 'DEF')
_L.define('dphys_rollout_fwd(Tensor z, Tensor? mu, Tensor controls, Tensor x0, Tensor xd0, Tensor R0, Tensor w0, Tensor pts, '
          'Tensor part_id, Tensor Iinv, float[] consts, int integrator, bool save_for_bwd) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)')
_L.define('dphys_rollout_bwd(Tensor z, Tensor? mu, Tensor controls, Tensor x_init, Tensor xd0, Tensor R0, Tensor w0, Tensor pts, '
          'Tensor part_id, Tensor Iinv, float[] consts, int integrator, Tensor Xraw, Tensor Xds, Tensor Rs, Tensor Om, Tensor rec, '
          'Tensor? gXs, Tensor? gXds, Tensor? gRs, Tensor? gOm, Tensor? gFs, Tensor? gFf) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)')
_L.define('bev_splat_plan(Tensor geom, float[] dx, float[] bx, int[] nx) -> Tensor')
_L.define('bev_splat_fwd(Tensor x, Tensor plan, int B, int n_per_sample, float[] dx, float[] bx, int[] nx) -> Tensor')
_L.define('bev_splat_bwd(Tensor grad, Tensor plan, int B, int n_per_sample, int C, float[] dx, float[] bx, int[] nx) -> Tensor')


def _sfx(dtype):
    if dtype == torch.float32:
        return 'f32'
    if dtype == torch.float64:
        return 'f64'
    raise TypeError(f'monoforce ops compute in float32 or float64, got {dtype}')


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _rollout_desc(z, controls, pts, Iinv, consts, integrator):
    assert len(consts) in (len(CONST_NAMES), len(CONST_NAMES) + 1), f'consts = {CONST_NAMES} (+ traj_sim_time)'
    c = dict(zip(CONST_NAMES, (float(v) for v in consts)))
    B, T = controls.shape[:2]
    assert z.dim() == 3 and z.shape[0] in (1, B), 'z must be [B,H,W] or [1,H,W] (one map shared by all rollouts)'
    d = _lib.MfRolloutDesc(B=B, T=T, N=pts.shape[0], H=z.shape[1], W=z.shape[2], n_tracks=2, integrator=int(integrator),
                           layout=_lib.MF_LAYOUT_TIME_MAJOR, map_shared=int(z.shape[0] == 1), math_mode=_lib.MF_MATH_FAST,
                           mass=c['mass'], gravity=c['gravity'], stiffness=c['stiffness'], damping=c['damping'], omega_max=c['omega_max'],
                           grid_res=c['grid_res'], d_max=c['d_max'], dt=c['dt'], robot_size_y=c['robot_size_y'])
    for i, v in enumerate(Iinv.detach().double().flatten().tolist()):      # 9 host scalars (a device read: build the op's inputs once)
        d.Iinv[i] = v
    return d


_TS = {}


def _time_grid(consts, T, dt, dev):
    """The reference's grid: linspace(0, T_sim, int(T_sim / dt))[:T] (dphysics.py:166-167, 581); T_sim defaults to T * dt."""
    step = float(consts[6])
    t_sim = float(consts[9]) if len(consts) > 9 else T * step
    key = (step, t_sim, T, dt, str(dev))
    if key not in _TS:       # built once per configuration: a host-to-device copy is not allowed while a stream is capturing
        _TS[key] = torch.linspace(0, t_sim, max(int(t_sim / step), T), dtype=dt)[:T].to(dev).contiguous()
    return _TS[key]


def _prep(z, mu, controls, pts, part_id, states=()):
    """Inputs in the kernels' dtype and layout, after the shape checks `DPhysics._make_desc` applies to the module path: both maps
    [1 or B, H, W] over the same H x W; when one is shared and the other per rollout the kernels index both at b*H*W, so the shared
    one is expanded for real (the backward sums its gradient back over the rollouts); every start-state tensor has the B rows of
    `controls`."""
    dt = z.dtype
    B = controls.shape[0]
    assert controls.dim() == 3 and controls.shape[2] == 2, f'controls must be [B,T,2], got {tuple(controls.shape)}'
    assert z.dim() == 3 and z.shape[0] in (1, B), f'z must be [B,H,W] or [1,H,W] (one map shared by all {B} rollouts), got {tuple(z.shape)}'
    if mu is not None:
        assert mu.dim() == 3 and tuple(mu.shape[1:]) == tuple(z.shape[1:]), \
            f'mu shape {tuple(mu.shape)} does not match the {z.shape[1]}x{z.shape[2]} height grid'
        assert mu.shape[0] in (1, B), f'mu batch {mu.shape[0]} is neither 1 (shared map) nor the {B} rollouts of controls'
        if mu.shape[0] != z.shape[0]:
            z, mu = z.expand(B, -1, -1), mu.expand(B, -1, -1)
    for name, t in states:
        assert t.shape[0] == B, f'{name} has {t.shape[0]} rows, controls {B} rollouts'
    assert pts.dim() == 2 and pts.shape[1] == 3 and part_id.shape[0] == pts.shape[0], 'pts must be [N,3] with one part id per point'
    cont = lambda t: None if t is None else t.to(dt).contiguous()  # noqa: E731
    return cont(z), cont(mu), cont(controls), cont(pts), part_id.to(torch.int32).contiguous()


@torch.library.impl(_L, 'dphys_rollout_fwd', 'CUDA')
def _rollout_fwd(z, mu, controls, x0, xd0, R0, w0, pts, part_id, Iinv, consts, integrator, save_for_bwd):
    dev, dt = z.device, z.dtype
    zc, muc, cc, pc, part = _prep(z, mu, controls, pts, part_id, (('x0', x0), ('xd0', xd0), ('R0', R0), ('w0', w0)))
    d = _rollout_desc(zc, cc, pc, Iinv, consts, integrator)
    if dt == torch.float64:
        d.math_mode = _lib.MF_MATH_EXACT
    B, T, N = d.B, d.T, d.N
    x0 = x0.to(dt).contiguous().clone()          # the kernel moves its z component onto the terrain: returned, not written in place
    with torch.cuda.device(dev):      # (the policy queries read the CU count of the CURRENT device)
        Np = _lib.lib().mf_rollout_force_stride(C.byref(d))
        # the component-parallel kernels' per-step record for the backward (MfRolloutFwdBufs.rec), where the library keeps one
        nrec = int(_lib.lib().mf_rollout_record_bytes(C.byref(d))) // 4 if (save_for_bwd and dt == torch.float32) else 0
    d.force_stride = Np
    new = lambda *tail: torch.empty(T, B, *tail, dtype=dt, device=dev)  # noqa: E731
    Xs, Xds, Rs, Om, Fs, Ff = new(3), new(3), new(3, 3), new(3), new(Np, 3), new(Np, 3)
    Xraw = new(3) if save_for_bwd else torch.empty(0, dtype=dt, device=dev)
    rec = torch.empty(nrec, dtype=dt, device=dev)
    ts = _time_grid(consts, T, dt, dev)
    bufs = _lib.MfRolloutFwdBufs(z=_lib.ptr(zc), mu=_lib.ptr(muc), controls=_lib.ptr(cc), ts=_lib.ptr(ts), points=_lib.ptr(pc),
                                 part=_lib.ptr(part), x0=_lib.ptr(x0), xd0=_lib.ptr(xd0.to(dt).contiguous()), R0=_lib.ptr(R0.to(dt).contiguous()),
                                 w0=_lib.ptr(w0.to(dt).contiguous()), Xs=_lib.ptr(Xs), Xds=_lib.ptr(Xds), Rs=_lib.ptr(Rs), Omegas=_lib.ptr(Om),
                                 Fs=_lib.ptr(Fs), Ff=_lib.ptr(Ff), Xraw=_lib.ptr(Xraw) if save_for_bwd else None,
                                 rec=_lib.ptr(rec) if nrec else None)
    with torch.cuda.device(dev), _timing.timed('rollout_fwd_kernel', dev):
        _lib.check(getattr(_lib.lib(), 'mf_rollout_fwd_' + _sfx(dt))(C.byref(d), C.byref(bufs), _stream(dev)), 'mf_rollout_fwd')
    tr = lambda t: t.transpose(0, 1)  # noqa: E731
    return tr(Xs), tr(Xds), tr(Rs), tr(Om), tr(Fs[:, :, :N]), tr(Ff[:, :, :N]), (tr(Xraw) if save_for_bwd else Xraw), x0, rec


@torch.library.impl(_L, 'dphys_rollout_bwd', 'CUDA')
def _rollout_bwd(z, mu, controls, x_init, xd0, R0, w0, pts, part_id, Iinv, consts, integrator, Xraw, Xds, Rs, Om, rec, gXs, gXds, gRs, gOm, gFs, gFf):
    dev, dt = z.device, z.dtype
    zc, muc, cc, pc, part = _prep(z, mu, controls, pts, part_id, (('x_init', x_init), ('xd0', xd0), ('R0', R0), ('w0', w0)))
    d = _rollout_desc(zc, cc, pc, Iinv, consts, integrator)
    if dt == torch.float64:
        d.math_mode = _lib.MF_MATH_EXACT
    B = d.B
    tm = lambda t: None if t is None else t.to(dt).transpose(0, 1).contiguous()  # noqa: E731    ([B,T,..] -> time-major rows)
    saved = [tm(t) for t in (Xraw, Xds, Rs, Om)]
    ups = [tm(t) for t in (gXs, gXds, gRs, gOm, gFs, gFf)]
    if d.map_shared:
        copies = grad_copies_for(B, int(d.N))
        d.grad_copies = copies
        n_maps = 2 if muc is not None else 1
        pool = grad_pool(_POOL_OWNER, n_maps, copies, zc[0].numel(), dt, dev)
        maps = pool.buf[:n_maps * copies * zc[0].numel()].view((n_maps, copies) + tuple(zc.shape[1:]))
        gz, gmu, zero_row = maps[0], (maps[1] if muc is not None else None), pool.buf[-16:]
    else:
        gz, gmu = torch.zeros_like(zc), (torch.zeros_like(muc) if muc is not None else None)
        zero_row = torch.zeros(16, dtype=dt, device=dev)
    gcontrols = torch.empty_like(cc)
    gx0, gxd0, gR0, gw0 = (torch.empty(B, 3, dtype=dt, device=dev), torch.empty(B, 3, dtype=dt, device=dev),
                           torch.empty(B, 3, 3, dtype=dt, device=dev), torch.empty(B, 3, dtype=dt, device=dev))
    ts = _time_grid(consts, d.T, dt, dev)
    bufs = _lib.MfRolloutBwdBufs(
        z=_lib.ptr(zc), mu=_lib.ptr(muc), controls=_lib.ptr(cc), ts=_lib.ptr(ts), points=_lib.ptr(pc), part=_lib.ptr(part),
        x_init=_lib.ptr(x_init.to(dt).contiguous()), xd0=_lib.ptr(xd0.to(dt).contiguous()), R0=_lib.ptr(R0.to(dt).contiguous()),
        w0=_lib.ptr(w0.to(dt).contiguous()), Xraw=_lib.ptr(saved[0]), Xds=_lib.ptr(saved[1]), Rs=_lib.ptr(saved[2]), Omegas=_lib.ptr(saved[3]),
        gXs=_lib.ptr(ups[0]), gXds=_lib.ptr(ups[1]), gRs=_lib.ptr(ups[2]), gOmegas=_lib.ptr(ups[3]), gFs=_lib.ptr(ups[4]), gFf=_lib.ptr(ups[5]),
        zeros=_lib.ptr(zero_row), gz=_lib.ptr(gz), gmu=_lib.ptr(gmu), gcontrols=_lib.ptr(gcontrols), gx0=_lib.ptr(gx0), gxd0=_lib.ptr(gxd0),
        gR0=_lib.ptr(gR0), gw0=_lib.ptr(gw0), rec=_lib.ptr(rec) if rec.numel() else None)
    with torch.cuda.device(dev), _timing.timed('rollout_bwd_kernel', dev):
        _lib.check(getattr(_lib.lib(), 'mf_rollout_bwd_' + _sfx(dt))(C.byref(d), C.byref(bufs), _stream(dev)), 'mf_rollout_bwd')
    if d.map_shared:
        summed = pool.reduce(zc.shape[1:])
        gz = summed[0].unsqueeze(0)
        gmu = summed[1].unsqueeze(0) if muc is not None else None
    # a map that came in as ONE shared map next to a per-rollout one was expanded in _prep: its gradient is the sum over the rollouts
    if gz.shape[0] != z.shape[0]:
        gz = gz.sum(0, keepdim=True)
    if gmu is not None and gmu.shape[0] != mu.shape[0]:
        gmu = gmu.sum(0, keepdim=True)
    if gmu is None:
        gmu = torch.zeros(0, dtype=dt, device=dev)
    return gz, gmu, gcontrols, gx0, gxd0, gR0, gw0


def _rollout_setup(ctx, inputs, output):
    z, mu, controls, x0, xd0, R0, w0, pts, part_id, Iinv, consts, integrator, save_for_bwd = inputs
    ctx.consts, ctx.integrator, ctx.has_mu = list(consts), integrator, mu is not None
    if not save_for_bwd:
        ctx.ok = False
        return
    ctx.ok = True
    Xs, Xds, Rs, Om, Fs, Ff, Xraw, x0s, rec = output
    ctx.save_for_backward(z, mu, controls, x0s, xd0, R0, w0, pts, part_id, Iinv, Xraw, Xds, Rs, Om, rec)
    ctx.set_materialize_grads(False)


def _rollout_backward(ctx, gXs, gXds, gRs, gOm, gFs, gFf, _gXraw, _gx0s, _grec):
    if not ctx.ok:
        raise RuntimeError('monoforce::dphys_rollout_fwd was called with save_for_bwd=False: no gradient available')
    z, mu, controls, x_init, xd0, R0, w0, pts, part_id, Iinv, Xraw, Xds, Rs, Om, rec = ctx.saved_tensors
    gz, gmu, gc, gx0, gxd0, gR0, gw0 = torch.ops.monoforce.dphys_rollout_bwd(
        z, mu, controls, x_init, xd0, R0, w0, pts, part_id, Iinv, ctx.consts, ctx.integrator, Xraw, Xds, Rs, Om, rec, gXs, gXds, gRs, gOm, gFs, gFf)
    return gz, (gmu if ctx.has_mu else None), gc, gx0, gxd0, gR0, gw0, None, None, None, None, None, None


torch.library.register_autograd('monoforce::dphys_rollout_fwd', _rollout_backward, setup_context=_rollout_setup, lib=_L)


@torch.library.register_fake('monoforce::dphys_rollout_fwd', lib=_L)
def _rollout_fwd_fake(z, mu, controls, x0, xd0, R0, w0, pts, part_id, Iinv, consts, integrator, save_for_bwd):
    B, T = controls.shape[:2]
    N = pts.shape[0]
    e = lambda *s: z.new_empty(s)  # noqa: E731
    return e(B, T, 3), e(B, T, 3), e(B, T, 3, 3), e(B, T, 3), e(B, T, N, 3), e(B, T, N, 3), (e(B, T, 3) if save_for_bwd else e(0)), e(B, 3), e(0)


@torch.library.register_fake('monoforce::dphys_rollout_bwd', lib=_L)
def _rollout_bwd_fake(z, mu, controls, x_init, xd0, R0, w0, pts, part_id, Iinv, consts, integrator, Xraw, Xds, Rs, Om, rec, gXs, gXds, gRs, gOm, gFs, gFf):
    B = controls.shape[0]
    return (torch.empty_like(z), torch.empty_like(mu) if mu is not None else z.new_empty(0), torch.empty_like(controls), z.new_empty(B, 3),
            z.new_empty(B, 3), z.new_empty(B, 3, 3), z.new_empty(B, 3))


# ---- BEV voxel pooling -------------------------------------------------------------------------------------------------
def _splat_desc(B, n_per_sample, Cc, dx, bx, nx):
    off = (torch.tensor(bx, dtype=torch.float32) - torch.tensor(dx, dtype=torch.float32) / 2.).tolist()      # float32, like lss.py:246
    return _lib.MfSplatDesc(B=B, n_per_sample=n_per_sample, C=Cc, nx=int(nx[0]), ny=int(nx[1]), nz=int(nx[2]),
                            off=(C.c_float * 3)(*off), dx=(C.c_float * 3)(*[float(v) for v in dx]))


@torch.library.impl(_L, 'bev_splat_plan', 'CUDA')
def _splat_plan(geom, dx, bx, nx):
    B = geom.shape[0]
    g = geom.detach().to(torch.float32).contiguous().view(-1, 3)
    d = _splat_desc(B, g.shape[0] // B, 1, dx, bx, nx)
    nbytes = _lib.lib().mf_bev_splat_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise RuntimeError('mf_bev_splat_workspace_bytes: ' + _lib.lib().mf_last_error().decode())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=geom.device)
    with torch.cuda.device(geom.device), _timing.timed('splat_prepare', geom.device):
        _lib.check(_lib.lib().mf_bev_splat_prepare(C.byref(d), _lib.ptr(g), _lib.ptr(ws), _stream(geom.device)), 'mf_bev_splat_prepare')
    return ws


@torch.library.impl(_L, 'bev_splat_fwd', 'CUDA')
def _splat_fwd(x, plan, B, n_per_sample, dx, bx, nx):
    Cc = x.shape[-1]
    xf = x.contiguous().view(-1, Cc)
    assert xf.shape[0] == B * n_per_sample, 'features and plan disagree on the number of points'
    d = _splat_desc(B, n_per_sample, Cc, dx, bx, nx)
    out = torch.empty(B, d.nz * Cc, d.nx, d.ny, dtype=xf.dtype, device=xf.device)
    with torch.cuda.device(xf.device), _timing.timed('splat_fwd_kernel', xf.device):
        _lib.check(getattr(_lib.lib(), 'mf_bev_splat_fwd_' + _sfx(xf.dtype))(C.byref(d), _lib.ptr(xf), _lib.ptr(plan), _lib.ptr(out), _stream(xf.device)),
                   'mf_bev_splat_fwd')
    return out


@torch.library.impl(_L, 'bev_splat_bwd', 'CUDA')
def _splat_bwd(grad, plan, B, n_per_sample, Cc, dx, bx, nx):
    g = grad.contiguous()
    d = _splat_desc(B, n_per_sample, Cc, dx, bx, nx)
    gx = torch.empty(B * n_per_sample, Cc, dtype=g.dtype, device=g.device)
    with torch.cuda.device(g.device), _timing.timed('splat_bwd_kernel', g.device):
        _lib.check(getattr(_lib.lib(), 'mf_bev_splat_bwd_' + _sfx(g.dtype))(C.byref(d), _lib.ptr(g), _lib.ptr(plan), _lib.ptr(gx), _stream(g.device)),
                   'mf_bev_splat_bwd')
    return gx


def _splat_setup(ctx, inputs, output):
    x, plan, B, n_per_sample, dx, bx, nx = inputs
    ctx.save_for_backward(plan)
    ctx.args = (B, n_per_sample, x.shape[-1], list(dx), list(bx), list(nx))
    ctx.x_shape = x.shape


def _splat_backward(ctx, grad):
    plan, = ctx.saved_tensors
    B, n, Cc, dx, bx, nx = ctx.args
    return torch.ops.monoforce.bev_splat_bwd(grad, plan, B, n, Cc, dx, bx, nx).view(ctx.x_shape), None, None, None, None, None, None


torch.library.register_autograd('monoforce::bev_splat_fwd', _splat_backward, setup_context=_splat_setup, lib=_L)


@torch.library.register_fake('monoforce::bev_splat_fwd', lib=_L)
def _splat_fwd_fake(x, plan, B, n_per_sample, dx, bx, nx):
    return x.new_empty(B, int(nx[2]) * x.shape[-1], int(nx[0]), int(nx[1]))


@torch.library.register_fake('monoforce::bev_splat_bwd', lib=_L)
def _splat_bwd_fake(grad, plan, B, n_per_sample, Cc, dx, bx, nx):
    return grad.new_empty(B * n_per_sample, Cc)


# ---- functional entries ------------------------------------------------------------------------------------------------------
def rollout(dphysics, z_grid, controls, state, friction=None):
    """`DPhysics.forward` through `torch.ops.monoforce.dphys_rollout_fwd` for a rigid body and a given start state:
    returns ((Xs, Xds, Rs, Omegas), (F_springs, F_frictions)); differentiable w.r.t. the maps, controls and start state."""
    cfg = dphysics.dphys_cfg
    dev, dt = z_grid.device, z_grid.dtype
    x0, xd0, R0, w0 = state
    consts = [float(cfg.robot_mass), float(cfg.gravity), float(dphysics.stiffness), float(dphysics.damping), float(cfg.grid_res),
              float(cfg.d_max), float(cfg.dt), float(cfg.omega_max), float(cfg.robot_size[1]), float(dphysics._ts_T)]
    pts = dphysics._points_dev(dev, dt)
    Iinv = torch.tensor(dphysics._iinv(dt), dtype=torch.float64).view(3, 3)
    need = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (z_grid, friction, controls, x0, xd0, R0, w0))
    outs = torch.ops.monoforce.dphys_rollout_fwd(z_grid, friction, controls, x0, xd0, R0, w0, pts, dphysics._part_dev(dev), Iinv, consts,
                                                 1 if cfg.use_odeint else 0, need)
    with torch.no_grad():
        x0.data[..., 2] = outs[7][..., 2].to(x0.dtype)       # the reference's in-place terrain snap of the caller's start position
    return tuple(outs[:4]), tuple(outs[4:6])


def splat(geom, x, dx, bx, nx):
    """`LiftSplatShoot.voxel_pooling(geom, x)` through `torch.ops.monoforce.bev_splat_*` (plan built per call)."""
    B = geom.shape[0]
    dxl, bxl, nxl = [float(v) for v in dx], [float(v) for v in bx], [int(v) for v in nx]
    plan = torch.ops.monoforce.bev_splat_plan(geom, dxl, bxl, nxl)
    n = geom.numel() // 3 // B
    return torch.ops.monoforce.bev_splat_fwd(x, plan, B, n, dxl, bxl, nxl)
