"""One rollout of a soak problem (written for mw 2006, SOAK_SEED0=2000: rollout 3's map gradient is 1.8 % off in float32 while its positions follow float64 to 1.5e-6 and the
oracle's own float32 gradient is stable under displacements and point orders.  Where does it come from?  (i) fast math or float32 itself: the
precise float32 kernels; (ii) when: the error against the horizon.)
    python tools/debug_soak_rollout.py <kind> <seed> <rollout>      (GPU; MONOFORCE_HIP_LIB selects an A/B build of the kernels)"""
import os, sys
os.environ.setdefault('OMP_NUM_THREADS', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(8)
from tests import soak_cases as sc, helpers as hp
from tests.test_rollout_gpu import make_dphysics
import tests.soak_cases as scm

kind, seed, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
c = sc.build(kind, seed)
n = c.sel.numel()
print({a: getattr(c, a) for a in ('B', 'T', 'H', 'res', 'N', 'integ', 'shared', 'all_outputs')}, getattr(c, 'where', None))
full_ctrl = c.ctrl
for precise in (False, True):
    real = scm.make_dphysics
    scm.make_dphysics = lambda *a, **kw: real(*a, **dict(kw, precise=precise))      # noqa: E731
    try:
        for T_ in (c.T, c.T // 2, c.T // 4, 3 * c.T // 4, 7 * c.T // 8):
            with sc.truncated(c, T_) as ct:
                e = sc.single_rollout_errors(ct, k)
            print('precise' if precise else 'fast   ', 'T', T_, {a: ('%.2e' % v[0], '%.2e' % v[1]) for a, v in e.items()}, flush=True)
    finally:
        scm.make_dphysics = real
        c.ctrl = full_ctrl
# where the gradient differs: the cells
g = sc.run_hip(c, rows_mask=sc.onehot(n, k))
o64 = sc.run_oracle(c, torch.float64, rows=torch.tensor([k]))
gh = g['gz'][k:k + 1] if g['gz'].shape[0] == n and n > 1 else g['gz']
d = (gh.double() - o64['gz']).abs()[0]
idx = torch.nonzero(d > 0.1 * d.max())
print('cells with > 10 % of the largest difference:', [(int(i), int(j), '%.3e' % float(d[i, j]), '%.3e' % float(o64['gz'][0, i, j])) for i, j in idx[:12]], 'largest float64 entry %.3e' % float(o64['gz'].abs().max()))
X = o64['Xs'][0]
P = torch.as_tensor(c.pts, dtype=torch.float64)
R = o64['Rs'][0]
x = X - R[:, :, 2] * sc.SINK
p = x.unsqueeze(1) + torch.einsum('tij,nj->tni', R, P)
u = (p[..., :2] + c.d_max) / c.res
print('cell coordinates of the contact points over the horizon: min %.4f max %.4f (map 0 .. %d)' % (float(u.min()), float(u.max()), c.H - 1))
