"""Trajectory shooting throughput: full-output rollout + torch costs vs states-only + torch costs vs the path-cost kernel."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing
from monoforce_amd.planner import TrajectoryShooter, sample_controls
DEV = 'cuda'
for B in [int(x) for x in os.environ.get('AB_B', '1024,4096,16384,65536').split(',')]:
    cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, 4, DEV, 1)
    zd, md = z.to(DEV), mu.to(DEV)
    c = sample_controls(B, cfg, DEV, torch.Generator(device=DEV).manual_seed(0))
    for cost in ('inclination', 'force'):
        res = {}
        for name, kw in (('fused', dict(fused=True)), ('unfused', dict(fused=False))):
            sh = TrajectoryShooter(dp, n_trajs=B, cost=cost, **kw)
            for _ in range(3): out = sh.shoot(zd, friction=md, controls=c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _timing.start(); e0.record()
            for _ in range(5): out = sh.shoot(zd, friction=md, controls=c)
            e1.record(); torch.cuda.synchronize()
            k = {n: float(np.mean(v)) for n, v in _timing.stop().items()}
            res[name] = dict(shoot_ms=round(e0.elapsed_time(e1) / 5, 3), kernel_ms=round(k['rollout_fwd_kernel'], 3), best=out['best'])
        print(json.dumps(dict(B=B, cost=cost, **res, rollout_steps_per_s_fused=round(B * 500 / res['fused']['shoot_ms'] * 1e3))), flush=True)
