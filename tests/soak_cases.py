"""The soak problems that landed outside the oracle-derived bar in round 5 (profiles/r5i_soak_win.txt: LDS-window problem 53; profiles/r5h_soak.txt:
multi-wave problems 118 and 146), rebuilt from their seeds with the generators of tools/soak_win.py and tools/soak_r5.py, so that
tests/test_soak_outliers_gpu.py can hold each one to its EXPLANATION instead of a text file.  Test infrastructure (imports oracle/).

    python -m tests.soak_cases <kind> <seed> <out.pt>      dumps the float32 HIP gradients of the case (the child process of the route-equality test:
                                                            MF_BWD_WIN / MF_MW_BWD are read once per process)"""
import os
import sys

import numpy as np
import torch

from oracle import dphysics_oracle as orc
from tests import helpers as hp
from tests.test_rollout_gpu import make_dphysics

DEV = 'cuda'
SINK = 40.0 * 9.81 / (5e4 + 1e-6)      # m g / (k + 1e-6): Xs = x + R[:, 2] * sink (dphysics.py:587-589), tradr mass, default stiffness


class Case:
    pass


def build(kind, seed):
    from monoforce_amd import synthetic as syn
    c = Case()
    c.kind, c.seed = kind, seed
    if kind == 'win':      # tools/soak_win.py
        rng = np.random.RandomState(9000 + seed)
        B = int(rng.choice([4608, 6144, 8192, 8192 + 512, 12288, 16384, 16384 + 37, 24576, 32768, 32768 + 4]))
        T = int(rng.randint(20, 121)); H = int(rng.choice([64, 128, 256, 512])); res = 12.8 / H
        N = int(rng.choice([3, 4])); integ = int(rng.randint(0, 2)); friction = bool(rng.randint(0, 4)); scattered = bool(rng.randint(0, 2))
        pts4, _ = syn.robot_points_4()
        pts = pts4[:N].copy(); masks = [pts[:, 1] > 0, pts[:, 1] <= 0]
        z = (syn.bump_terrain(syn.bump_params(seed + 3), 6.4, res) * float(rng.choice([0.3, 1.0]))).unsqueeze(0)
        mu = syn.wave_friction(6.4, res).unsqueeze(0) if friction else None
        ctrl = syn.const_controls(B, T, seed=seed)
        sub = 24
        sel = torch.cat([torch.arange(0, B, B // (sub - 3))[:sub - 3], torch.arange(B - 3, B)])
        state = None
        if scattered:
            g = torch.Generator().manual_seed(seed)
            x0 = torch.zeros(B, 3); x0[:, :2] = (torch.rand(B, 2, generator=g) - 0.5) * 12.4
            yaw = torch.rand(B, generator=g) * 6.2831853
            R0 = torch.zeros(B, 3, 3); R0[:, 0, 0] = yaw.cos(); R0[:, 0, 1] = -yaw.sin(); R0[:, 1, 0] = yaw.sin(); R0[:, 1, 1] = yaw.cos(); R0[:, 2, 2] = 1.0
            xd0 = torch.zeros(B, 3); xd0[:, 0] = ctrl[:, 0, 0] * yaw.cos(); xd0[:, 1] = ctrl[:, 0, 0] * yaw.sin()
            w0 = torch.zeros(B, 3); w0[:, 2] = ctrl[:, 0, 1]
            state = (x0, xd0, R0, w0)
        c.B, c.T, c.H, c.res, c.d_max, c.N, c.integ, c.shared = B, T, H, res, 6.4, N, integ, True
        c.pts, c.masks, c.z, c.mu, c.ctrl, c.sel, c.state = pts, masks, z, mu, ctrl, sel, state
        c.wts = syn.probe_weights((sel.numel(), T, 3), phase=0.1 * seed)
        c.all_outputs = False
    elif kind == 'cp':     # tools/soak_r5.py, the component-parallel generator (tests/test_random_shapes_gpu.py::_cp_case): <= 4 points, the
        # N-point body run with the 4-point body's inertia and track width on both sides
        from tests.test_random_shapes_gpu import _cp_case
        info, pts, masks, z, mu, ctrl, state, d_max = _cp_case(seed)
        c.B, c.T, c.H, c.res, c.d_max, c.N, c.integ, c.shared = info['B'], info['T'], info['H'], info['res'], d_max, info['N'], info['integ'], info['shared']
        c.pts, c.masks, c.z, c.mu, c.ctrl, c.sel, c.state = pts, masks, z, mu, ctrl, torch.arange(info['B']), state
        c.loss_kind, c.all_outputs, c.wts = info['loss'], info['loss'] != 1, None
    else:                  # tools/soak_r5.py, the multi-wave generator
        rng = np.random.RandomState(seed)
        N = int(rng.choice([5, 7, 8, 9, 16, 17, 32, 33, 50, 64, 65, 100, 128, 129, 175, 223, 256, 257, 300]))
        B = int(rng.randint(1, 41)); T = int(rng.choice([2, 3, 5, 17, 40, 80, 120]))
        nt = int(rng.choice([2, 4])); integ = int(rng.randint(0, 2)); shared = bool(rng.randint(0, 2)); use_mu = bool(rng.randint(0, 3))
        rs_ = float(rng.choice([0.05, 0.1])); d_max = 3.2; xs_only = bool(rng.randint(0, 2))
        pts, masks = syn.robot_points_box(N, seed=seed, n_tracks=nt)
        nb = 1 if shared else B
        z = torch.stack([syn.bump_terrain(syn.bump_params(seed + b, smooth=bool(rng.randint(0, 2))), d_max, rs_, torch.float64) * 0.3 for b in range(nb)]).float()
        mu = torch.stack([syn.wave_friction(d_max, rs_, 0.5, 1.0, 1.0 + 0.1 * b, 0.8, torch.float64) for b in range(nb)]).float()
        ctrl = syn.varying_controls(B, T, seed=seed, dtype=torch.float64).float()
        where = rng.choice(['centre', 'edge', 'off'])
        x0 = torch.zeros(B, 3); x0[:, 0] = {'centre': 0.0, 'edge': d_max - 0.3, 'off': d_max + 0.5}[where]; x0[:, 1] = torch.from_numpy(rng.uniform(-1, 1, B)).float()
        yaw = torch.from_numpy(rng.uniform(-3.1, 3.1, B)).float()
        R0 = torch.zeros(B, 3, 3); R0[:, 0, 0] = yaw.cos(); R0[:, 0, 1] = -yaw.sin(); R0[:, 1, 0] = yaw.sin(); R0[:, 1, 1] = yaw.cos(); R0[:, 2, 2] = 1
        xd0 = torch.stack([yaw.cos(), yaw.sin(), torch.zeros(B)], 1) * 0.8
        c.B, c.T, c.H, c.res, c.d_max, c.N, c.integ, c.shared = B, T, z.shape[-1], rs_, d_max, N, integ, shared
        c.pts, c.masks, c.z, c.mu, c.ctrl, c.sel = pts, masks, z, (mu if use_mu else None), ctrl, torch.arange(B)
        c.state = (x0, xd0, R0, torch.zeros(B, 3))
        c.all_outputs, c.where = not xs_only, str(where)
        c.wts = None
    c.spec = hp.spec_from(c.pts, c.masks, c.integ, c.res, c.d_max)
    if kind == 'cp':
        pts4, _ = syn.robot_points_4()
        c.spec.robot_size_y = float(pts4[:, 1].max() - pts4[:, 1].min())
    return c


class _body4:
    """cp problems: the oracle with the 4-point body's inertia (what the HIP side of the soak runs the N-point body with)."""
    def __init__(self, c):
        self.on = c.kind == 'cp'

    def __enter__(self):
        if self.on:
            from monoforce_amd import synthetic as syn
            self.keep = orc.point_inertia
            P4 = torch.as_tensor(syn.robot_points_4()[0], dtype=torch.float32)
            orc.point_inertia = lambda mass, P, _pi=self.keep: _pi(mass, P4.to(P).unsqueeze(0))

    def __exit__(self, *exc):
        if self.on:
            orc.point_inertia = self.keep


def _dphysics(c, points_per_lane, precise):
    kw = dict(precise=True) if precise else {}
    if c.kind != 'cp':
        return make_dphysics(c.pts, c.masks, c.integ, c.res, c.d_max, points_per_lane=points_per_lane, **kw)
    from monoforce_amd import synthetic as syn
    pts4, _ = syn.robot_points_4()
    m4 = [pts4[:, 1] > 0, pts4[:, 1] <= 0]
    lanes = points_per_lane or (1 if os.environ.get('MF_SOAK_CP_LANES') == '0' else 16)      # (the other float32 route: one point per lane)
    base = make_dphysics(pts4, m4, c.integ, c.res, c.d_max)
    dp = make_dphysics(pts4, m4, c.integ, c.res, c.d_max, points_per_lane=lanes, **kw)
    dp.dphys_cfg.robot_points = torch.as_tensor(c.pts)
    dp.dphys_cfg.driving_parts = [torch.as_tensor(m) for m in c.masks]
    dp.x_points = dp.dphys_cfg.robot_points.unsqueeze(0).to(dp.device)
    dp._cache = {('iinv', dt_): base._iinv(dt_) for dt_ in (torch.float32, torch.float64)}
    return dp


def _loss(c, outs, dt, dev, rows_mask, widx=None):
    """The soak's probe loss over the selected rollouts; `rows_mask` [n] switches rollouts off; `widx`: positions (in the full selection) of
    the rollouts at hand, so that a subset is weighted like the same rollouts of the full problem."""
    from monoforce_amd import synthetic as syn
    m = rows_mask.to(dt).to(dev)
    n_full = c.sel.numel()
    pick = (lambda w: w) if widx is None else (lambda w: w[widx.to(w.device)])
    if c.kind == 'win':
        return (outs[0] * pick(c.wts.to(dt).to(dev)) * m.view(-1, 1, 1)).sum()
    if c.kind == 'cp':      # tools/soak_r5.py loss_of: all outputs (hp.probe_loss) / positions only / positions + spring forces
        W = lambda o, ph: pick(syn.probe_weights((n_full,) + tuple(o.shape[1:]), phase=ph, dtype=dt).to(o.device)) * m.view(-1, *([1] * (o.dim() - 1)))      # noqa: E731
        if c.loss_kind == 0:
            return sum((o * W(o, 0.5 + i)).sum() * s_ for i, (o, s_) in enumerate(zip(outs, [1.0, 1.0, 1.0, 1.0, 1e-3, 1e-3])))
        loss = (outs[0] * W(outs[0], 0.4)).sum()
        return loss + 1e-3 * (outs[4] * W(outs[4], 1.4)).sum() if c.loss_kind == 2 else loss
    if not c.all_outputs:
        X = outs[0][:, ::3]
        return (X * pick(syn.probe_weights((n_full,) + tuple(X.shape[1:]), 0.3, dtype=dt).to(dev)) * m.view(-1, 1, 1)).sum()
    scales = [1.0, 1.0, 1.0, 1.0, 1e-3, 1e-3]
    loss = 0
    for i, (o, s) in enumerate(zip(outs, scales)):
        mm = m.view(-1, *([1] * (o.dim() - 1)))
        loss = loss + (o * pick(syn.probe_weights((n_full,) + tuple(o.shape[1:]), phase=0.5 + i, dtype=dt).to(o.device)) * mm).sum() * s
    return loss


def _maps(c, t, n, rows):
    if t is None:
        return None
    if t.shape[0] == 1:
        return t.expand(n, -1, -1) if c.kind != 'win' else t
    return t if rows is None else t[rows]


def run_hip(c, dt=torch.float32, rows=None, rows_mask=None, points_per_lane=0, precise=False):
    """Gradients (z, mu, controls of the selected rollouts) and outputs of the HIP route.  `rows` (positions in the selection): run ONLY these
    rollouts as their own small batch (other kernels: what the float64 check uses)."""
    from monoforce_amd import _timing
    dp = _dphysics(c, points_per_lane, precise)
    dp.dphys_cfg.traj_sim_time = 5.0
    widx = rows                                   # positions in the selection (weights); `rows` below: rollout indices
    rows = None if rows is None else c.sel[rows]
    idx = c.sel if rows is None else rows
    ctrl = c.ctrl if rows is None else c.ctrl[rows]
    n = ctrl.shape[0]
    leaf = lambda t: t.to(dt).to(DEV).detach().clone().requires_grad_(True)      # noqa: E731
    per_rollout_maps = c.z.shape[0] > 1
    zsrc = c.z if (rows is None or not per_rollout_maps) else c.z[rows]
    msrc = None if c.mu is None else (c.mu if (rows is None or not per_rollout_maps) else c.mu[rows])
    zd, md, cd = leaf(zsrc), (leaf(msrc) if msrc is not None else None), leaf(ctrl)
    ex = lambda m: None if m is None else (m.expand(n, -1, -1) if (m.shape[0] == 1 and c.kind != 'win') else m)      # noqa: E731
    st = None
    if c.state is not None:
        st = tuple((t if rows is None else t[rows]).clone().to(dt).to(DEV) for t in c.state)
    _timing.start()
    states, forces = dp(ex(zd), cd, friction=ex(md), state=st)
    outs = list(states) + list(forces)
    sel_local = idx.to(DEV) if rows is None else torch.arange(n, device=DEV)
    mask = torch.ones(sel_local.numel()) if rows_mask is None else rows_mask
    _loss(c, [o[sel_local] for o in outs], dt, DEV, mask, widx).backward()
    name = _timing.launches().get('rollout_bwd_kernel', '?').split(' grid')[0]
    _timing.stop()
    return dict(gz=zd.grad.cpu(), gmu=(md.grad.cpu() if md is not None else None), gc=cd.grad[sel_local].cpu(),
                Xs=outs[0][sel_local].detach().cpu(), Rs=outs[2][sel_local].detach().cpu(), kernel=name)


def run_oracle(c, dt, rows=None, rows_mask=None):
    widx = rows
    idx = c.sel if rows is None else c.sel[rows]
    n = idx.numel()
    per_rollout_maps = c.z.shape[0] > 1
    leaf = lambda t: t.detach().clone().to(dt).requires_grad_(True)      # noqa: E731  (a copy: .to(float32) of a float32 tensor is the tensor itself)
    zc = leaf(c.z if not per_rollout_maps else c.z[idx])
    mc = None if c.mu is None else leaf(c.mu if not per_rollout_maps else c.mu[idx])
    cc = leaf(c.ctrl[idx])
    st = tuple(t[idx].clone().to(dt) for t in c.state) if c.state is not None else None
    ex = lambda m: None if m is None else (m.expand(n, -1, -1) if m.shape[0] == 1 else m)      # noqa: E731
    with _body4(c):
        states, forces = orc.rollout(c.spec, ex(zc), cc, state=st, friction=ex(mc))
    outs = list(states) + list(forces)
    mask = torch.ones(n) if rows_mask is None else rows_mask
    _loss(c, outs, dt, 'cpu', mask, widx).backward()
    return dict(gz=zc.grad, gmu=(mc.grad if mc is not None else None), gc=cc.grad, Xs=outs[0].detach(), Xds=outs[1].detach(), Rs=outs[2].detach(),
                Om=outs[3].detach())


def onehot(n, k):
    m = torch.zeros(n); m[k] = 1.0
    return m


def single_rollout_errors(c, k, g_hip=None, precise=False, with_diff=False):
    """Rollout k's OWN contribution to every gradient (the loss restricted to it): HIP float32 -- the SAME kernel and launch as the full problem,
    only the loss's weights change -- and the oracle's float32, each against the oracle's float64 (run on that rollout alone), relative to the
    largest float64 entry.  Returns {key: (e_hip, e_o32)}."""
    n = c.sel.numel()
    g = g_hip if g_hip is not None else run_hip(c, rows_mask=onehot(n, k), precise=precise)
    rows = torch.tensor([k])
    o64, o32 = run_oracle(c, torch.float64, rows=rows), run_oracle(c, torch.float32, rows=rows)
    out = {}
    for key in ('gz', 'gmu', 'gc'):
        if o64[key] is None:
            continue
        gh = g[key]
        gh = gh[k:k + 1] if gh.shape[0] == n and n > 1 else gh      # per-rollout maps / control rows: this rollout's; a shared map: the whole
        scale = float(o64[key].abs().max())
        out[key] = (float((gh.double() - o64[key]).abs().max()) / scale, float((o32[key].double() - o64[key]).abs().max()) / scale)
    if with_diff:      # ... and WHERE the map gradient differs
        return out, (g['gz'][k:k + 1] if g['gz'].shape[0] == n and n > 1 else g['gz']).double() - o64['gz']
    return out


def kink_rows(c, k, o64=None, edge_ulps=4.0, clamp_rel=1e-4):
    """Output rows of rollout k (float64 oracle) at which the contact model -- evaluated AT that state, for the step that follows -- sits on a
    kink within float32 resolution: a contact point within `edge_ulps` float32 ulps of a cell edge (the interpolant is continuous there, its
    slopes and the cells a gradient lands in are not), an unclamped spring force / friction force / angular acceleration within `clamp_rel`
    of its clamp (dphysics.py:233,250-251,257: the value is continuous, the derivative switches between 1 and 0), or the spring-damper term
    k dh + b v_n within `clamp_rel` of zero (the friction force scales with |F_n|, dphysics.py:238: its derivative changes sign there).
    [(row, kind, margin)]."""
    dt = torch.float64
    if o64 is None:
        o64 = run_oracle(c, dt, rows=torch.tensor([k]))
        Xs, Xds, Rs, Om = (o64[a][0] for a in ('Xs', 'Xds', 'Rs', 'Om'))
    else:
        Xs, Xds, Rs, Om = (o64[a][k] for a in ('Xs', 'Xds', 'Rs', 'Om'))
    spec = c.spec
    idx = c.sel[torch.tensor([k])]
    per = c.z.shape[0] > 1
    z = (c.z[idx] if per else c.z).to(dt)
    mu = torch.ones_like(z) if c.mu is None else (c.mu[idx] if per else c.mu).to(dt)
    ctrl = c.ctrl[idx].to(dt)[0]
    P = spec.points.to(dt).unsqueeze(0)
    with _body4(c):
        Iinv = torch.linalg.inv(orc.point_inertia(spec.mass, P))
    mg = spec.mass * spec.gravity
    x = Xs - Rs[:, :, 2] * SINK
    T_ = Xs.shape[0]
    tq = torch.arange(T_)
    tc = (tq + 1).clamp_max(ctrl.shape[0] - 1) if c.integ == 0 else tq      # dynamics(): the step that starts from row t is step t + 1
    p = P @ Rs.transpose(1, 2) + x.unsqueeze(1)                            # [T, N, 3]: the T rows as a batch
    r = p - x.unsqueeze(1)
    vp = Xds.unsqueeze(1) + torch.linalg.cross(Om.unsqueeze(1).expand_as(r), r)
    zq, n = orc.sample_grid(z.expand(T_, -1, -1), p[..., 0], p[..., 1], spec.d_max, spec.grid_res, normals=True)
    muq = orc.sample_grid(mu.expand(T_, -1, -1), p[..., 0], p[..., 1], spec.d_max, spec.grid_res).unsqueeze(-1)
    dh = p[..., 2:3] - zq.unsqueeze(-1)
    cc = torch.sigmoid(-10.0 * dh)
    vn = (vp * n).sum(2, keepdim=True)
    A_ = spec.stiffness * dh + spec.damping * vn                       # the spring-damper term: |F_n| = |A| c / sum(c) |n| has a kink at A = 0
    F1 = -torch.mul(A_, n) * cc / cc.sum(1, keepdim=True)
    Fs = torch.clamp(F1, -mg, mg)
    e = orc.unit(Rs[..., 0])
    tv = orc.track_speeds(ctrl[tc, 0], ctrl[tc, 1], spec.robot_size_y, len(spec.driving_parts))
    cmd = torch.zeros_like(vp)
    for j in range(len(spec.driving_parts)):
        cmd[:, spec.driving_parts[j]] = (tv[:, j].unsqueeze(1) * e).unsqueeze(1)
    slip = muq * (cmd - vp)
    Gf = torch.norm(Fs, dim=2).unsqueeze(2) * (slip - (slip * n).sum(2, keepdim=True) * n)
    wd = (Iinv @ torch.sum(torch.linalg.cross(r, Fs + torch.clamp(Gf, -mg, mg)), 1).unsqueeze(2)).squeeze(2)
    u = (p[..., :2] + spec.d_max) / spec.grid_res
    fr = u - torch.floor(u)
    m_edge = (torch.minimum(fr, 1 - fr) / (2.0 ** -23 * u.abs().clamp_min(1.0))).flatten(1).amin(1)      # in float32 ulps of the coordinate
    m_F = ((F1.abs() - mg).abs() / mg).flatten(1).amin(1)
    m_Ff = ((Gf.abs() - mg).abs() / mg).flatten(1).amin(1)
    m_wd = ((wd.abs() - spec.omega_max).abs() / spec.omega_max).amin(1)
    m_A = (A_.abs() / ((spec.stiffness * dh).abs() + (spec.damping * vn).abs()).clamp_min(1e-30)).flatten(1).amin(1)
    out = []
    for t in range(T_):
        for kind_, mval, lim in (('cell edge (float32 ulps)', m_edge[t], edge_ulps), ('spring-force clamp', m_F[t], clamp_rel),
                                 ('friction-force clamp', m_Ff[t], clamp_rel), ('angular-acceleration clamp', m_wd[t], clamp_rel),
                                 ('sign of the spring-damper term (|F_n| kink)', m_A[t], clamp_rel)):
            if float(mval) <= lim:
                out.append((t, kind_, float(mval)))
    return out


class truncated:
    """The same problem stopped after `T_` steps (controls and, for the window problems, the probe weights cut)."""
    def __init__(self, c, T_):
        self.c, self.T_ = c, int(T_)

    def __enter__(self):
        c = self.c
        self.keep = (c.ctrl, c.wts, c.T)
        c.ctrl, c.T = c.ctrl[:, :self.T_].contiguous(), self.T_
        if c.wts is not None:
            c.wts = c.wts[:, :self.T_].contiguous()
        return c

    def __exit__(self, *exc):
        self.c.ctrl, self.c.wts, self.c.T = self.keep


if __name__ == '__main__':
    kind, seed, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    r = run_hip(build(kind, seed))
    torch.save(r, out)
