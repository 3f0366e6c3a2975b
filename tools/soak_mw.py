"""Random problems for the record-reading backward (rollout_bwd_mw_kernel.h) against the general recomputing kernel
(`points_per_lane=4`): body size 5..300, 1..40 rollouts, 2..120 steps, both integrators, shared / per-rollout maps, with / without
a friction map, resolution 0.05 / 0.1, starts in the map, near its edge or off it, all-outputs / positions-only losses.
    python tools/soak_mw.py [n_seeds] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monoforce_amd import synthetic as syn
from tests import helpers as hp
from tests.test_rollout_gpu import make_dphysics
DEV = 'cuda'
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
worst, bad = 0.0, []
for seed in range(first, first + n_seeds):
    rng = np.random.RandomState(seed)
    N = int(rng.choice([5, 7, 8, 9, 16, 17, 32, 33, 50, 64, 65, 100, 128, 129, 175, 223, 256, 257, 300]))
    B = int(rng.randint(1, 41)); T = int(rng.choice([2, 3, 5, 17, 40, 80, 120]))
    nt = int(rng.choice([2, 4])); integ = int(rng.randint(0, 2)); shared = bool(rng.randint(0, 2)); use_mu = bool(rng.randint(0, 3))
    res = float(rng.choice([0.05, 0.1])); d_max = 3.2; xs_only = bool(rng.randint(0, 2))
    pts, masks = syn.robot_points_box(N, seed=seed, n_tracks=nt)
    nb = 1 if shared else B
    z = torch.stack([syn.bump_terrain(syn.bump_params(seed + b, smooth=bool(rng.randint(0, 2))), d_max, res, torch.float64) * 0.3 for b in range(nb)]).float()
    mu = torch.stack([syn.wave_friction(d_max, res, 0.5, 1.0, 1.0 + 0.1 * b, 0.8, torch.float64) for b in range(nb)]).float()
    ctrl = syn.varying_controls(B, T, seed=seed, dtype=torch.float64).float()
    where = rng.choice(['centre', 'edge', 'off'])
    x0 = torch.zeros(B, 3); x0[:, 0] = {'centre': 0.0, 'edge': d_max - 0.3, 'off': d_max + 0.5}[where]; x0[:, 1] = torch.from_numpy(rng.uniform(-1, 1, B)).float()
    yaw = torch.from_numpy(rng.uniform(-3.1, 3.1, B)).float()
    R0 = torch.zeros(B, 3, 3); R0[:, 0, 0] = yaw.cos(); R0[:, 0, 1] = -yaw.sin(); R0[:, 1, 0] = yaw.sin(); R0[:, 1, 1] = yaw.cos(); R0[:, 2, 2] = 1
    xd0 = torch.stack([yaw.cos(), yaw.sin(), torch.zeros(B)], 1) * 0.8
    state = (x0, xd0, R0, torch.zeros(B, 3))

    def run(ppl):
        dp = make_dphysics(pts, masks, integ, res, d_max, points_per_lane=ppl)
        zl, cl = z.clone().to(DEV).requires_grad_(True), ctrl.clone().to(DEV).requires_grad_(True)
        ml = mu.clone().to(DEV).requires_grad_(True) if use_mu else None
        ex = lambda m: None if m is None else (m.expand(B, -1, -1) if m.shape[0] == 1 else m)  # noqa: E731
        st = [t.clone().to(DEV) for t in state]
        for t in st[1:]:
            t.requires_grad_(True)
        so, fo = dp(ex(zl), cl, state=tuple(st), friction=ex(ml))
        outs = list(so) + list(fo)
        loss = (outs[0][:, ::3] * syn.probe_weights(outs[0][:, ::3].shape, 0.3).to(DEV)).sum() if xs_only else hp.probe_loss(outs, torch.float32)
        loss.backward()
        return [g.grad.cpu() for g in [zl, cl] + ([ml] if use_mu else []) + st[1:]]

    ga, gb = run(0), run(4)
    errs = [hp.rel_err(a, b) for a, b in zip(ga, gb)]
    fin = all(bool(torch.isfinite(a).all()) for a in ga)
    e = max(errs)
    worst = max(worst, e)
    if e > 2e-3 or not fin:
        bad.append((seed, dict(N=N, B=B, T=T, nt=nt, integ=integ, shared=shared, mu=use_mu, res=res, where=str(where), xs=xs_only), [round(v, 5) for v in errs], fin))
        print('BAD', bad[-1], flush=True)
print(f'seeds {first}..{first + n_seeds - 1}: worst relative difference {worst:.2e}, {len(bad)} above 2e-3', flush=True)
