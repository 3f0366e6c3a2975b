"""CPU oracle for the DPhysics rollout  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
The shipped path (`monoforce_amd`) never does: it calls the HIP library or fails loudly.

What it is: a functional torch-CPU restatement of the reference algorithm (float32 or float64 following
the dtype of its inputs, differentiable through torch autograd), written against
`/root/reference/monoforce/src/monoforce/models/traj_predictor/dphysics.py` (cited per function as `dphysics.py:L`).
It is pinned against the real reference by `tests/golden/*.npz`, which `tests/golden/gen_golden.py`
produced in the build container by importing the reference itself (the reference ships no tests or
golden vectors of its own).

Third-party arithmetic: the reference's default integrator calls `torchdiffeq.odeint(method='euler')`
(`dphysics.py:510-511`; torchdiffeq==0.2.3 per `monoforce/docker/requirements.txt:22`, NOT vendored in
/root/reference and not installed here).  Its published fixed-grid Euler is restated in `rollout()`:
`y[n+1] = y[n] + (t[n+1]-t[n]) * f(t[n], y[n])`, outputs at the grid points, `y[0]` = initial state.
That boundary is "parity unpinned" in the strict sense (no reference-side test pins it); the golden
vectors for that mode were produced with the same 10-line restatement plugged in as the `odeint` symbol.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch

DYNAMICS = 0       # reference `dynamics()` loop: semi-implicit Euler + Rodrigues (dphysics.py:467-497)
ODEINT_EULER = 1   # reference default: torchdiffeq fixed-grid explicit Euler (dphysics.py:499-528)


@dataclass
class RolloutSpec:
    """Constants the rollout consumes (fields of DPhysConfig, dphys_config.py:77-153)."""
    points: torch.Tensor                 # [N,3] body-frame contact points (cfg.robot_points)
    driving_parts: List[torch.Tensor]    # list of bool masks [N] (2 or 4 tracks)
    robot_size_y: float                  # Ly = robot_size[1]
    mass: float = 40.0
    gravity: float = 9.81
    stiffness: float = 50_000.0
    damping: Optional[float] = None      # default sqrt(4 m k) (dphys_config.py:143)
    omega_max: float = 2.0
    grid_res: float = 0.1
    d_max: float = 6.4
    dt: float = 0.01
    traj_sim_time: float = 5.0
    integrator: int = ODEINT_EULER
    joint_positions: Optional[list] = None   # [[x,y,z]] * n_parts (cfg.joint_positions.values()), only with joint angles

    def __post_init__(self):
        if self.damping is None:
            self.damping = math.sqrt(4 * self.mass * self.stiffness)


def unit(v, eps=1e-6):
    """v / max(|v|, eps) along the last axis (dphysics.py:7-19)."""
    return v / torch.clamp(torch.norm(v, dim=-1, keepdim=True), min=eps)


def hat(w):
    """[w]x for w[B,3] (dphysics.py:22-40)."""
    z = torch.zeros_like(w[:, 0])
    return torch.stack([torch.stack([z, -w[:, 2], w[:, 1]], -1),
                        torch.stack([w[:, 2], z, -w[:, 0]], -1),
                        torch.stack([-w[:, 1], w[:, 0], z], -1)], 1)


def point_inertia(mass, pts):
    """Inertia of N equal point masses about the body origin, pts[B,N,3] -> [B,3,3] (dphysics.py:107-141)."""
    mp = mass / pts.shape[1]
    x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]
    xx = torch.sum(mp * (y ** 2 + z ** 2), 1)
    yy = torch.sum(mp * (x ** 2 + z ** 2), 1)
    zz = torch.sum(mp * (x ** 2 + y ** 2), 1)
    xy = -torch.sum(mp * x * y, 1)
    xz = -torch.sum(mp * x * z, 1)
    yz = -torch.sum(mp * y * z, 1)
    return torch.stack([torch.stack([xx, xy, xz], 1), torch.stack([xy, yy, yz], 1), torch.stack([xz, yz, zz], 1)], 1)


def track_speeds(v, w, Ly, n_tracks):
    """(v,w) -> per-track speeds [B,n_tracks] (dphysics.py:75-104). NB the 2- and 4-track forms round differently."""
    if n_tracks == 2:
        return torch.stack([v - w * (Ly / 2.0), v + w * (Ly / 2.0)], -1)
    if n_tracks == 4:
        lo, hi = v - w * Ly / 2.0, v + w * Ly / 2.0
        return torch.stack([lo, hi, lo, hi], -1)
    raise ValueError('n_tracks must be 2 or 4')


def sample_grid(grid, qx, qy, d_max, res, normals=False):
    """The reference's `interpolate_grid` (dphysics.py:385-455), bug-for-bug:
    trunc-toward-zero cell index, flat index clamped to [0, HW-1] (wraps rows), grid axis 0 = x with
    stride H, and the x-fraction weighting the +y neighbour (and vice versa)."""
    B, H, W = grid.shape
    flat = grid.reshape(B, -1)
    ux, uy = (qx + d_max) / res, (qy + d_max) / res
    ix, iy = ux.long(), uy.long()
    fx, fy = ux - ix.to(ux.dtype), uy - iy.to(uy.dtype)
    last = H * W - 1
    i_c = torch.clamp(iy + H * ix, 0, last)
    i_f = torch.clamp(iy + H * (ix + 1), 0, last)
    i_l = torch.clamp((iy + 1) + H * ix, 0, last)
    i_fl = torch.clamp((iy + 1) + H * (ix + 1), 0, last)
    zc, zf, zl, zfl = flat.gather(1, i_c), flat.gather(1, i_f), flat.gather(1, i_l), flat.gather(1, i_fl)
    z = (1 - fx) * (1 - fy) * zc + (1 - fx) * fy * zf + fx * (1 - fy) * zl + fx * fy * zfl
    if not normals:
        return z
    gx, gy = (zf - zc) / res, (zl - zc) / res
    n = unit(torch.stack([-gx, -gy, torch.ones_like(gx)], -1))
    return z, n


def articulate(spec, P, joint_angles_t):
    """`update_joints` (dphysics.py:326-358): rotate each driving part about the y-axis through its joint position.
    P[1,N,3], joint_angles_t[B,4] -> points[B,N,3].  The reference applies this only for robot == 'marv' and angles != 0."""
    B = joint_angles_t.shape[0]
    pts = P.repeat(B, 1, 1)
    for i, mask in enumerate(spec.driving_parts):
        xyz = torch.as_tensor(spec.joint_positions[i], dtype=pts.dtype).view(1, 1, 3)
        a = joint_angles_t[:, i]
        o, z = torch.ones_like(a), torch.zeros_like(a)
        Ry = torch.stack([torch.cos(a), z, torch.sin(a), z, o, z, -torch.sin(a), z, torch.cos(a)], 1).view(B, 3, 3)
        pts[:, mask] = (pts[:, mask] - xyz) @ Ry.transpose(1, 2) + xyz
    return pts


def rhs(spec, Iinv, P, part_id, z_grid, mu_grid, ctrl, x, xd, R, w):
    """One evaluation of `forward_kinematics` (dphysics.py:172-272) with joint angles == 0.

    P[1,N,3] body points, part_id[N] (index of the LAST driving mask containing the point, -1 if none),
    ctrl[B,2].  Returns (xdd, dR, wd), (F_spring[B,N,3], F_friction[B,N,3]).
    """
    m, g = spec.mass, spec.gravity
    p = P @ R.transpose(1, 2) + x.unsqueeze(1)                                   # :200
    r = p - x.unsqueeze(1)
    vp = xd.unsqueeze(1) + torch.linalg.cross(w.unsqueeze(1).expand_as(r), r)    # :204
    zq, n = sample_grid(z_grid, p[..., 0], p[..., 1], spec.d_max, spec.grid_res, normals=True)   # :211
    mu = sample_grid(mu_grid, p[..., 0], p[..., 1], spec.d_max, spec.grid_res).unsqueeze(-1)     # :216
    dh = p[..., 2:3] - zq.unsqueeze(-1)                                          # :220
    c = torch.sigmoid(-10.0 * dh)                                                # :223
    vn = (vp * n).sum(2, keepdim=True)                                           # :228
    Fs = -torch.mul(spec.stiffness * dh + spec.damping * vn, n)                  # :230
    Fs = torch.mul(Fs, c) / torch.sum(c, 1, keepdim=True)                        # :231-232
    Fs = torch.clamp(Fs, -m * g, m * g)                                          # :233
    e = unit(R[..., 0])                                                          # :237 (first column of R)
    Nn = torch.norm(Fs, dim=2)                                                   # :238
    tv = track_speeds(ctrl[:, 0], ctrl[:, 1], spec.robot_size_y, len(spec.driving_parts))        # :239
    cmd = torch.zeros_like(vp)                                                   # :242-246
    for j in range(len(spec.driving_parts)):
        cmd[:, spec.driving_parts[j]] = (tv[:, j].unsqueeze(1) * e).unsqueeze(1)
    slip = mu * (cmd - vp)                                                       # :247
    slip_t = slip - (slip * n).sum(2, keepdim=True) * n                          # :248-249
    Ff = torch.clamp(Nn.unsqueeze(2) * slip_t, -m * g, m * g)                    # :250-251
    tau = torch.sum(torch.linalg.cross(r, Fs + Ff), 1)                           # :255
    wd = (Iinv @ tau.unsqueeze(2)).squeeze(2)                                    # :256
    wd = torch.clamp(wd, -spec.omega_max, spec.omega_max)                        # :257
    dR = hat(w) @ R                                                              # :258-259
    Fg = m * g * torch.tensor([0.0, 0.0, -1.0], dtype=x.dtype).unsqueeze(0)      # :264
    xdd = (Fg + Fs.sum(1) + Ff.sum(1)) / m                                       # :265-266
    return (xdd, dR, wd), (Fs, Ff)


def rodrigues_step(R, w, dt, eps=1e-6):
    """R @ (I + K sin(th dt) + K^2 (1 - cos(th dt))), K = [w]x / max(|w|, eps) (dphysics.py:291-324)."""
    th = torch.norm(w, dim=-1, keepdim=True).unsqueeze(-1)
    K = hat(w) / torch.clamp(th, min=eps)
    I = torch.eye(3, dtype=R.dtype)
    return R @ (I + K * torch.sin(th * dt) + K @ K * (1 - torch.cos(th * dt)))


def time_grid(spec, n_controls, dtype):
    """`ts = linspace(0, T, int(T/dt))[:N_ts]`, N_ts = min(int(T/dt), controls.shape[1]) (dphysics.py:166-167,573,581)."""
    n_full = int(spec.traj_sim_time / spec.dt)
    ts = torch.linspace(0, spec.traj_sim_time, n_full, dtype=dtype)
    return ts[:min(n_full, n_controls)]


def rollout(spec: RolloutSpec, z_grid, controls, state=None, friction=None, ts=None, joint_angles=None):
    """`DPhysics.dphysics` (dphysics.py:530-594) for joint angles == 0.

    z_grid[B,H,W], controls[B,T,2], optional state=(x[B,3], xd[B,3], R[B,3,3], w[B,3]), friction[B,H,W].
    Like the reference it overwrites `state[0][:, 2]` in place with the terrain height under the robot.
    Returns ((Xs, Xds, Rs, Omegas), (F_springs, F_frictions)) laid out [B,T,...].
    `ts` overrides the time grid (the reference's linspace is built in the default dtype at construction).
    """
    dtype = z_grid.dtype
    B = z_grid.shape[0]
    P = spec.points.to(dtype).unsqueeze(0)
    N = P.shape[1]
    part_id = torch.full((N,), -1, dtype=torch.long)
    for j, mk in enumerate(spec.driving_parts):
        part_id[mk] = j
    Iinv = torch.linalg.inv(point_inertia(spec.mass, P))                          # :152-153 / :196-197

    if state is None:                                                             # :554-559
        x = torch.zeros(B, 3, dtype=dtype)
        xd = torch.zeros(B, 3, dtype=dtype); xd[:, 0] = controls[:, 0, 0]
        R = torch.eye(3, dtype=dtype).repeat(B, 1, 1)
        w = torch.zeros(B, 3, dtype=dtype); w[:, 2] = controls[:, 0, 1]
    else:
        x, xd, R, w = state
    if friction is None:                                                          # :562 (cfg.friction == ones)
        friction = torch.ones_like(z_grid)
    p0 = P.repeat(B, 1, 1) @ R.transpose(1, 2) + x.unsqueeze(1)                   # :567-571
    x[..., 2:3] = sample_grid(z_grid, p0[..., 0], p0[..., 1], spec.d_max, spec.grid_res).mean(1, keepdim=True)

    if ts is None:
        ts = time_grid(spec, controls.shape[1], dtype)
    T = ts.shape[0]
    assert controls.shape == (B, T, 2), f'Controls shape {tuple(controls.shape)} != {(B, T, 2)}'   # :575

    def body(n):
        # points and inverse inertia of step n: constants, or re-articulated flippers (dphysics.py:192-197)
        if joint_angles is None:
            return P, Iinv
        Pn = articulate(spec, P, joint_angles[:, n])
        return Pn, torch.linalg.inv(point_inertia(spec.mass, Pn))

    out = [[] for _ in range(6)]
    if spec.integrator == DYNAMICS:                                               # :467-497, :274-288
        h = spec.dt
        for n in range(T):
            Pn, In = body(n)
            (xdd, _, wd), (Fs, Ff) = rhs(spec, In, Pn, part_id, z_grid, friction, controls[:, n], x, xd, R, w)
            xd = xd + xdd * h
            x = x + xd * h
            w = w + wd * h
            R = rodrigues_step(R, w, h)
            for lst, v in zip(out, (x, xd, R, w, Fs, Ff)):
                lst.append(v)
    elif spec.integrator == ODEINT_EULER:                                         # :499-528 + torchdiffeq euler
        y = (x, xd, R, w, torch.zeros(B, N, 3, dtype=dtype), torch.zeros(B, N, 3, dtype=dtype))
        for lst, v in zip(out, y):
            lst.append(v)
        for n in range(T - 1):
            h = ts[n + 1] - ts[n]
            Pn, In = body(n)
            (xdd, dR, wd), (Fs, Ff) = rhs(spec, In, Pn, part_id, z_grid, friction, controls[:, n], *y[:4])
            f = (y[1], xdd, dR, wd, Fs, Ff)
            y = tuple(a + h * b for a, b in zip(y, f))
            for lst, v in zip(out, y):
                lst.append(v)
    else:
        raise ValueError(f'Unknown integrator: {spec.integrator}')
    Xs, Xds, Rs, Om, Fsp, Ffr = [torch.stack(l, 1) for l in out]
    sink = spec.mass * spec.gravity / (spec.stiffness + 1e-6)                     # :587
    Xs = Xs + Rs[:, :, :3, 2] * sink                                              # :589
    return (Xs, Xds, Rs, Om), (Fsp, Ffr)
