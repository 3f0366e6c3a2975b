"""Margins to every kink of the contact model along the float64 trajectory of ONE rollout of a soak problem: the clamps on the spring force,
the friction force and the angular acceleration (dphysics.py:233,250-251,257), the cell edges, the normalisations' floors.  CPU only (the
oracle).   python tools/debug_soak_margins.py <kind> <seed> <rollout>"""
import os, sys
os.environ.setdefault('OMP_NUM_THREADS', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(8)
from oracle import dphysics_oracle as orc
from tests import soak_cases as sc

kind, seed, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
c = sc.build(kind, seed)
dt = torch.float64
spec = c.spec
idx = c.sel[torch.tensor([k])]
per = c.z.shape[0] > 1
z = (c.z[idx] if per else c.z).to(dt)
mu = None if c.mu is None else (c.mu[idx] if per else c.mu).to(dt)
ctrl = c.ctrl[idx].to(dt)
st = tuple(t[idx].clone().to(dt) for t in c.state) if c.state is not None else None
(Xs, Xds, Rs, Om), _ = orc.rollout(spec, z, ctrl, state=tuple(t.clone() for t in st) if st else None, friction=mu)
P = spec.points.to(dt).unsqueeze(0)
N = P.shape[1]
part_id = torch.full((N,), -1, dtype=torch.long)
for j, mk in enumerate(spec.driving_parts):
    part_id[mk] = j
Iinv = torch.linalg.inv(orc.point_inertia(spec.mass, P))
m, g = spec.mass, spec.gravity
mug = torch.ones_like(z) if mu is None else mu
T = Xs.shape[1]
rows = []
for t in range(T):
    if c.integ == 0:      # dynamics(): step t starts from output row t - 1 (the given state for t = 0)
        if t == 0:
            x, xd, R, w = (s.clone() for s in st)
            x[:, 2] = Xs[:, 0, 2] * 0 + x[:, 2]
            continue      # (the snapped initial height is not reconstructed here)
        R, xd, w = Rs[:, t - 1], Xds[:, t - 1], Om[:, t - 1]
        x = Xs[:, t - 1] - R[:, :, 2] * sc.SINK
    else:
        R, xd, w = Rs[:, t], Xds[:, t], Om[:, t]
        x = Xs[:, t] - R[:, :, 2] * sc.SINK
    p = P @ R.transpose(1, 2) + x.unsqueeze(1)
    r = p - x.unsqueeze(1)
    vp = xd.unsqueeze(1) + torch.linalg.cross(w.unsqueeze(1).expand_as(r), r)
    zq, n = orc.sample_grid(z, p[..., 0], p[..., 1], spec.d_max, spec.grid_res, normals=True)
    muq = orc.sample_grid(mug, p[..., 0], p[..., 1], spec.d_max, spec.grid_res).unsqueeze(-1)
    dh = p[..., 2:3] - zq.unsqueeze(-1)
    cc = torch.sigmoid(-10.0 * dh)
    vn = (vp * n).sum(2, keepdim=True)
    F1 = -torch.mul(spec.stiffness * dh + spec.damping * vn, n) * cc / cc.sum(1, keepdim=True)
    Fs = torch.clamp(F1, -m * g, m * g)
    e = orc.unit(R[..., 0])
    Nn = torch.norm(Fs, dim=2)
    tv = orc.track_speeds(ctrl[:, t, 0], ctrl[:, t, 1], spec.robot_size_y, len(spec.driving_parts))
    cmd = torch.zeros_like(vp)
    for j in range(len(spec.driving_parts)):
        cmd[:, spec.driving_parts[j]] = (tv[:, j].unsqueeze(1) * e).unsqueeze(1)
    slip = muq * (cmd - vp)
    slip_t = slip - (slip * n).sum(2, keepdim=True) * n
    Gf = Nn.unsqueeze(2) * slip_t
    Ff = torch.clamp(Gf, -m * g, m * g)
    tau = torch.sum(torch.linalg.cross(r, Fs + Ff), 1)
    wd = (Iinv @ tau.unsqueeze(2)).squeeze(2)
    u = (p[..., :2] + spec.d_max) / spec.grid_res
    fr = u - torch.trunc(u)
    rows.append(dict(t=t, F=float(((F1.abs() - m * g).abs() / (m * g)).min()), Ff=float(((Gf.abs() - m * g).abs() / (m * g)).min()),
                     wd=float(((wd.abs() - spec.omega_max).abs() / spec.omega_max).min()), edge=float(torch.minimum(fr.abs(), (1 - fr).abs()).min()),
                     th=float(w.norm()), clampF=int((F1.abs() > m * g).sum()), clampFf=int((Gf.abs() > m * g).sum()), clampwd=int((wd.abs() > spec.omega_max).sum()),
                     umax=float(u.max()), wdmax=float(wd.abs().max())))
for key in ('F', 'Ff', 'wd', 'edge'):
    best = sorted(rows, key=lambda r: r[key])[:3]
    print('smallest relative margin to', key, [(r['t'], '%.2e' % r[key]) for r in best])
print('steps with an active clamp  F:', [r['t'] for r in rows if r['clampF']][:40], ' Ff:', [r['t'] for r in rows if r['clampFf']][:40], ' wd:', [r['t'] for r in rows if r['clampwd']][:40])
print('|w| min %.2e at t=%d' % min((r['th'], r['t']) for r in rows), ' u max %.3f' % max(r['umax'] for r in rows), ' (map 0 .. %d)' % (c.H - 1))
