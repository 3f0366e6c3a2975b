"""Host-side profile of a BASELINE configs[0] step (ONE rollout x 200 steps, forward + backward): that step is bound by the
Python / launch path, not by its kernels.  python tools/profile_c1_host.py [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_problem
from monoforce_amd import synthetic as syn
from monoforce_amd.train import TerrainFitProblem
dev = torch.device('cuda', 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg, dp, pts, masks, z, mu, ctrl = build_problem(1, 200, 4, dev, 1, seed=0, grid_res=0.1)
prob = TerrainFitProblem(dp, syn.bump_terrain(syn.bump_params(100), 6.4, 0.1).to(dev), mu.to(dev), ctrl.to(dev))
zl, ml = z.to(dev).clone().requires_grad_(True), mu.to(dev).clone().requires_grad_(True)
for _ in range(20):
    prob.step(zl, ml)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    prob.step(zl, ml)
t_launch = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'{steps} steps: host launch path {t_launch / steps * 1e3:.3f} ms/step, with the GPU drained {t_all / steps * 1e3:.3f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    prob.step(zl, ml)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
