"""BEV voxel pooling on the HIP kernels: `voxel_pooling(geom, x, dx, bx, nx)` == `LiftSplatShoot.voxel_pooling`
(`/root/reference/monoforce/src/monoforce/models/terrain_encoder/lss.py:238-280`) with exact per-voxel sums.

`SplatPlan` is the geometry-only part (voxel keys + CSR lists); it is built once per `geom` and shared by forward and
backward, and can be reused across steps while the camera calibration / augmentation is unchanged.
"""
import ctypes as C

import torch

from . import _lib, _timing


import os as _os
_HOST_INVERSE = bool(int(_os.environ.get('MF_SPLAT_HOST_INVERSE', '0')))      # round 4's plan route for camera rigs (torch.inverse on the device)


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def grid_host(dx, bx, nx):
    """(off, dx, n) of the BEV grid as host values (a device -> host read: a stream sync, not capturable in a hipGraph --
    callers that build plans per frame read it once and pass it as `grid=`)."""
    dxc, bxc = dx.detach().float().cpu(), bx.detach().float().cpu()
    off = bxc - dxc / 2.                                        # float32, like lss.py:246
    return off.tolist(), dxc.tolist(), [int(v) for v in nx.detach().cpu().tolist()]


class SplatPlan:
    def __init__(self, geom, dx, bx, nx, _cameras=None, grid=None, workspace=None):
        """geom [B, ..., 3] float32 on the GPU; dx, bx (float32 tensors), nx (int64 tensor) as produced by gen_dx_bx.
        `workspace`: a uint8 tensor of at least mf_bev_splat_workspace_bytes() to build the plan in (a caller that rebuilds the plan
        every step reuses one buffer: stream order keeps step i's kernels ahead of step i + 1's rebuild)."""
        if _cameras is None:
            _lib.require_hip_tensor(geom, 'geom')
            B = geom.shape[0]
            g = geom.detach().to(torch.float32).contiguous().view(-1, 3)
            device, n_per_sample = geom.device, g.shape[0] // B
        else:
            frustum, cams, B = _cameras      # cams: [B*N,24] coefficient rows, or the five raw calibration tensors (the rig entry point)
            rig = isinstance(cams, (tuple, list))
            _lib.require_hip_tensor(cams[0] if rig else cams, 'camera models')
            device, n_per_sample = (cams[0] if rig else cams).device, ((cams[0] if rig else cams).shape[0] // B) * frustum.shape[0]
        off, dxl, n = grid if grid is not None else grid_host(dx, bx, nx)
        self.B, self.n_per_sample = B, n_per_sample
        self.nx, self.ny, self.nz = n
        self.device = device
        self._desc_args = dict(B=B, n_per_sample=self.n_per_sample, nx=n[0], ny=n[1], nz=n[2],
                               off=(C.c_float * 3)(*off), dx=(C.c_float * 3)(*dxl))
        d = self.desc(1)
        nbytes = _lib.lib().mf_bev_splat_workspace_bytes(C.byref(d))
        if nbytes == 0:
            raise RuntimeError('mf_bev_splat_workspace_bytes: ' + _lib.lib().mf_last_error().decode())
        if workspace is not None and workspace.numel() >= nbytes and workspace.device == device and workspace.dtype == torch.uint8:
            self.workspace = workspace
        else:
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
        with torch.cuda.device(device), _timing.timed('splat_prepare', device):
            if _cameras is None:
                _lib.check(_lib.lib().mf_bev_splat_prepare(C.byref(d), _lib.ptr(g), _lib.ptr(self.workspace), _stream_ptr(device)),
                           'mf_bev_splat_prepare')
                self._keepalive = g
            elif rig:
                _lib.check(_lib.lib().mf_bev_splat_prepare_rig(C.byref(d), _lib.ptr(frustum), C.c_int32(frustum.shape[0]), *(_lib.ptr(t) for t in cams),
                                                               _lib.ptr(self.workspace), _stream_ptr(device)), 'mf_bev_splat_prepare_rig')
                self._keepalive = (frustum, cams)
            else:
                _lib.check(_lib.lib().mf_bev_splat_prepare_cameras(C.byref(d), _lib.ptr(frustum), C.c_int32(frustum.shape[0]),
                                                                   _lib.ptr(cams), _lib.ptr(self.workspace), _stream_ptr(device)),
                           'mf_bev_splat_prepare_cameras')
                self._keepalive = (frustum, cams)

    @classmethod
    def from_cameras(cls, frustum, rots, trans, intrins, post_rots, post_trans, dx, bx, nx, grid=None, host_inverse=None, workspace=None):
        """The plan of `SplatPlan(get_geometry(rots, trans, intrins, post_rots, post_trans), ...)` without the geometry
        tensor: frustum [D,fH,fW,3] (create_frustum), camera models [B,N,3(,3)] as LiftSplatShoot.forward takes them.
        Default (round 5): `mf_bev_splat_prepare_rig` -- the five tensors go to the key kernel as they are, which inverts post_rots and
        intrins itself: no `torch.inverse` (two LU launches + a host synchronisation), no concatenations, nothing on the host but one
        C call; a per-sample augmentation (new tensors every step) costs the plan's five launches.  `host_inverse=True` (or
        MF_SPLAT_HOST_INVERSE=1): round 4's route, torch.inverse on the device and `mf_bev_splat_prepare_cameras` -- torch's own LU
        bits (A/B runs, parity tests of the two)."""
        B, N = trans.shape[:2]
        f32 = lambda t: t.detach().to(torch.float32)
        fr = f32(frustum).contiguous().view(-1, 3)
        if host_inverse is None:
            host_inverse = _HOST_INVERSE
        if not host_inverse:
            rig = tuple(f32(t).reshape(B * N, -1).contiguous() for t in (rots, trans, intrins, post_rots, post_trans))
            return cls(None, dx, bx, nx, _cameras=(fr, rig, B), grid=grid, workspace=workspace)
        # both 3x3 inverses of get_geometry (lss.py:212, 218) in ONE batched call: the batched LU treats every matrix on its own, so
        # the results are the reference's, bit for bit, at half the launches
        inv = torch.inverse(torch.cat((f32(post_rots).reshape(B * N, 3, 3), f32(intrins).reshape(B * N, 3, 3))))
        cams = torch.cat((f32(post_trans).reshape(B * N, 3), inv[:B * N].reshape(B * N, 9),
                          f32(rots).reshape(B * N, 3, 3).matmul(inv[B * N:]).reshape(B * N, 9), f32(trans).reshape(B * N, 3)), 1).contiguous()
        return cls(None, dx, bx, nx, _cameras=(fr, cams, B), grid=grid, workspace=workspace)

    def keys(self):
        """Linear voxel id of every point ([B * n_per_sample] int32, -1 = dropped): the first array of the workspace."""
        return self.workspace[:4 * self.B * self.n_per_sample].view(torch.int32)

    def desc(self, C_):
        return _lib.MfSplatDesc(C=C_, **self._desc_args)


class _Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan):
        Cc = x.shape[-1]
        xf = x.contiguous().view(-1, Cc)
        assert xf.shape[0] == plan.B * plan.n_per_sample, 'features and geometry disagree on the number of points'
        sfx = {torch.float32: 'f32', torch.float64: 'f64'}[xf.dtype]
        out = torch.empty(plan.B, plan.nz * Cc, plan.nx, plan.ny, dtype=xf.dtype, device=xf.device)
        d = plan.desc(Cc)
        with torch.cuda.device(xf.device), _timing.timed('splat_fwd_kernel', xf.device):
            _lib.check(getattr(_lib.lib(), 'mf_bev_splat_fwd_' + sfx)(C.byref(d), _lib.ptr(xf), _lib.ptr(plan.workspace),
                                                                       _lib.ptr(out), _stream_ptr(xf.device)), 'mf_bev_splat_fwd')
        ctx.plan, ctx.x_shape, ctx.sfx = plan, x.shape, sfx
        return out

    @staticmethod
    def backward(ctx, gout):
        plan = ctx.plan
        Cc = ctx.x_shape[-1]
        g = gout.contiguous()
        gx = torch.empty(plan.B * plan.n_per_sample, Cc, dtype=g.dtype, device=g.device)
        d = plan.desc(Cc)
        with torch.cuda.device(g.device), _timing.timed('splat_bwd_kernel', g.device):
            _lib.check(getattr(_lib.lib(), 'mf_bev_splat_bwd_' + ctx.sfx)(C.byref(d), _lib.ptr(g), _lib.ptr(plan.workspace),
                                                                           _lib.ptr(gx), _stream_ptr(g.device)), 'mf_bev_splat_bwd')
        return gx.view(ctx.x_shape), None


class _LiftPool(torch.autograd.Function):
    """`mf_bev_lift_splat_*`: the lift `softmax(depth) (x) context` (lss.py:63-71) fused into the voxel pooling (lss.py:238-280);
    the [B,N,D,fH,fW,C] tensor of lifted features (30.9 MB per sample at config-4 shapes) is never built."""

    @staticmethod
    def forward(ctx, depth, context, plan):
        BN, D, fH, fW = depth.shape
        Cc = context.shape[1]
        assert context.shape == (BN, Cc, fH, fW) and BN * D * fH * fW == plan.B * plan.n_per_sample, \
            'depth / context and the geometry disagree on the number of frustum points'
        sfx = {torch.float32: 'f32', torch.float64: 'f64'}[depth.dtype]
        dep = depth.contiguous()
        ctxT = context.to(depth.dtype).permute(0, 2, 3, 1).contiguous()          # pixel-major rows [BN, fH, fW, C] (0.5 MB per sample)
        d = plan.desc(Cc)
        d.lift_D, d.lift_hw = D, fH * fW
        out = torch.empty(plan.B, plan.nz * Cc, plan.nx, plan.ny, dtype=dep.dtype, device=dep.device)
        with torch.cuda.device(dep.device), _timing.timed('lift_splat_fwd_kernel', dep.device):
            _lib.check(getattr(_lib.lib(), 'mf_bev_lift_splat_fwd_' + sfx)(C.byref(d), _lib.ptr(dep), _lib.ptr(ctxT), _lib.ptr(plan.workspace),
                                                                            _lib.ptr(out), _stream_ptr(dep.device)), 'mf_bev_lift_splat_fwd')
        ctx.save_for_backward(dep, ctxT)
        ctx.plan, ctx.d, ctx.sfx = plan, d, sfx
        return out

    @staticmethod
    def backward(ctx, gout):
        dep, ctxT = ctx.saved_tensors
        plan, d = ctx.plan, ctx.d
        g = gout.contiguous()
        rows = torch.empty(plan.B * plan.nz * plan.nx * plan.ny, d.C, dtype=g.dtype, device=g.device)     # voxel-major gradient rows
        g_dep, g_ctx = torch.empty_like(dep), torch.empty_like(ctxT)
        with torch.cuda.device(g.device), _timing.timed('lift_splat_bwd_kernel', g.device):
            _lib.check(getattr(_lib.lib(), 'mf_bev_lift_splat_bwd_' + ctx.sfx)(
                C.byref(d), _lib.ptr(dep), _lib.ptr(ctxT), _lib.ptr(plan.workspace), _lib.ptr(g), _lib.ptr(rows), _lib.ptr(g_dep),
                _lib.ptr(g_ctx), _stream_ptr(g.device)), 'mf_bev_lift_splat_bwd')
        return g_dep, g_ctx.permute(0, 3, 1, 2), None


def lift_voxel_pooling(geom, depth, context, dx, bx, nx, plan=None):
    """geom [B,N,D,fH,fW,3], depth [B*N,D,fH,fW], context [B*N,C,fH,fW] -> [B, C*nz, nx, ny]: == voxel_pooling(geom, lift)."""
    _lib.require_hip_tensor(depth, 'depth')
    if plan is None:
        plan = SplatPlan(geom, dx, bx, nx)
    return _LiftPool.apply(depth, context, plan)


def voxel_pooling(geom, x, dx, bx, nx, plan=None):
    """geom [B,N,D,H,W,3], x [B,N,D,H,W,C] -> [B, C*nz, nx, ny] (lss.py:238-280).  No gradient to geom, as in the reference."""
    _lib.require_hip_tensor(x, 'x')
    if plan is None:
        plan = SplatPlan(geom, dx, bx, nx)
    return _Pool.apply(x, plan)
