"""Why does the forward INSIDE a fit step cost more than the plain forward at saturated batches (r6a sweep: 16 384 rollouts 0.56 ms in the step,
0.34 ms alone)?  The same kernel family four ways: all six outputs (no_grad) | states only (no_grad) | states + Xraw for a backward, forwards
back to back | the forward of whole fit steps.  HIP events around the C-ABI launch; run it under `rocprofv3 --kernel-trace --stats` for the
kernel durations proper.   AB_B=16384,32768 python tools/ab_step_fwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing, synthetic as syn
from monoforce_amd.train import TerrainFitProblem
DEV = 'cuda'
for B in [int(x) for x in os.environ.get('AB_B', '16384,32768').split(',')]:
    cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, 4, DEV, int(os.environ.get('AB_INTEG', '1')))
    cd = ctrl.to(DEV)
    zd, md = z.to(DEV).unsqueeze(0), mu.to(DEV).unsqueeze(0)
    zl, ml = z.to(DEV).clone().requires_grad_(True), mu.to(DEV).clone().requires_grad_(True)
    prob = TerrainFitProblem(dp, syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(DEV), mu.to(DEV), cd)

    def timed(fn, n=4):
        fn(); fn()
        _timing.start()
        for _ in range(n):
            fn()
        inst = _timing.launches().get('rollout_fwd_kernel', '')
        k = {a: round(float(np.mean(v)), 4) for a, v in _timing.stop().items()}
        return k, inst[:110]

    def plain():
        with torch.no_grad():
            dp(zd, cd, friction=md)

    def states():
        dp.return_forces = False
        try:
            with torch.no_grad():
                dp(zd, cd, friction=md)
        finally:
            dp.return_forces = True

    def grad_fwd():      # the autograd forward of the fit step alone (states + Xraw), no backward
        dp.return_forces = False
        try:
            dp(zl.unsqueeze(0), cd, friction=ml.unsqueeze(0))
        finally:
            dp.return_forces = True

    for name, fn in (('all outputs, no_grad', plain), ('states only, no_grad', states), ('states + Xraw (autograd forward alone)', grad_fwd),
                     ('whole fit step', lambda: prob.step(zl, ml, eager=True)), ('all outputs again', plain)):
        k, inst = timed(fn)
        print(f'B={B} {name:42s} {k}  {inst}', flush=True)
    # the whole step end to end, replayed as one hipGraph: HIP events around 8 replays
    probg = TerrainFitProblem(dp, syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(DEV), mu.to(DEV), cd, graph=True)
    for _ in range(3):
        probg.step(zl, ml)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        probg.step(zl, ml)
    e1.record()
    torch.cuda.synchronize()
    print(f'B={B} step replayed as one hipGraph: {e0.elapsed_time(e1) / 8:.4f} ms', flush=True)
    del prob, probg, dp
    torch.cuda.empty_cache()
