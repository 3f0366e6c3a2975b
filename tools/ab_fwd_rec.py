"""Forward kernel with / without the per-step record at B = 256 / 1024 (fused-loss train step, as bench.py's c3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing, synthetic as syn
from monoforce_amd.train import TerrainFitProblem
for B in [int(x) for x in os.environ.get('AB_B', '256,1024,2048').split(',')]:
    _, dp, _, _, z, mu, cs = build_problem(B, 500, 4, 'cuda', 1, seed=0)
    cs = cs.cuda()
    prob = TerrainFitProblem(dp, syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).cuda(), mu.cuda(), cs)
    zl, ml = z.cuda().clone().requires_grad_(True), mu.cuda().clone().requires_grad_(True)
    for _ in range(3): prob.step(zl, ml)
    _timing.start()
    for _ in range(8): prob.step(zl, ml)
    k = {n: float(np.mean(v)) for n, v in _timing.stop().items()}
    print('B', B, {n: round(v, 4) for n, v in k.items() if 'rollout' in n}, flush=True)
