"""A/B of the saturated backward of <= 4-point bodies: the record-reading scan at four lanes per rollout
(MF_MW_SMALL=1: recording forward + rollout_bwd_mw_kernel<float, 4, ...>) against the recomputing kernels (component-parallel early
recompute up to 8192 rollouts, one point per lane with carry-over beyond).  One process per setting (the library reads the switch once):

    MF_MW_SMALL=0 python tools/ab_mw_small.py 8192 16384 32768 ; MF_MW_SMALL=1 python tools/ab_mw_small.py 8192 16384 32768

Prints per batch: kernels that ran, forward / backward kernel ms (HIP events), the step's gradient norms (the two routes must agree)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_problem  # noqa: E402
from monoforce_amd import _timing, synthetic as syn  # noqa: E402
from monoforce_amd.train import TerrainFitProblem  # noqa: E402

dev = torch.device('cuda', 0)
T = int(os.environ.get('AB_T', '500'))
integ = int(os.environ.get('AB_INTEG', '1'))
for Bs in [int(a) for a in sys.argv[1:]] or [8192, 16384]:
    _, dps, _, _, z, mu, cs = build_problem(Bs, T, 4, dev, integ, seed=0)
    cs = cs.to(dev)
    prob = TerrainFitProblem(dps, syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(dev), mu.to(dev), cs)
    zl, ml = z.to(dev).clone().requires_grad_(True), mu.to(dev).clone().requires_grad_(True)
    for _ in range(2):
        loss = prob.step(zl, ml)
    _timing.start()
    for _ in range(4):
        loss = prob.step(zl, ml)
    launches = _timing.launches()
    k = {n: float(np.mean(v)) for n, v in _timing.stop().items()}
    print(json.dumps({'B': Bs, 'MF_MW_SMALL': os.environ.get('MF_MW_SMALL', '0'), 'ms': k, 'loss': float(loss),
                      'gz_norm': float(zl.grad.norm()), 'gmu_norm': float(ml.grad.norm()), 'gz_absmax': float(zl.grad.abs().max()),
                      'kernels': launches}))
    del prob, dps
    torch.cuda.empty_cache()
