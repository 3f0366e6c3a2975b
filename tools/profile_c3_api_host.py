"""Host-side (cProfile) view of the drop-in route `c3_api` (bench.py::api_workload): DPhysics.forward + monoforce.losses.physics_loss +
loss.backward() at the BASELINE shape, launch by launch -- that route is bound by Python and launch calls, not by its kernels.
python tools/profile_c3_api_host.py [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_problem
from monoforce_amd import synthetic as syn
from monoforce.losses import physics_loss
dev = torch.device('cuda', 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, T = 1024, 500
cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, T, 4, dev, 1, seed=0)
cd = ctrl.to(dev)
with torch.no_grad():
    (Xg, Xdg, Rg, Og), _ = dp(syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(dev).unsqueeze(0), cd, friction=mu.to(dev).unsqueeze(0))
full_ts = torch.linspace(0, cfg.traj_sim_time, int(cfg.traj_sim_time / cfg.dt), device=dev)[:T]
sel = torch.arange(9, T, 10, device=dev)
pred_ts, gt_ts = full_ts.unsqueeze(0).expand(B, -1), full_ts[sel].unsqueeze(0).expand(B, -1).contiguous()
states_gt = [t[:, sel].contiguous() for t in (Xg, Xdg, Rg, Og)]
zl, ml = z.to(dev).clone().requires_grad_(True), mu.to(dev).clone().requires_grad_(True)


def step():
    zl.grad = ml.grad = None
    states, forces = dp(z_grid=zl.unsqueeze(0), controls=cd, friction=ml.unsqueeze(0))
    loss = physics_loss(states_pred=states, states_gt=states_gt, pred_ts=pred_ts, gt_ts=gt_ts, gamma=0.9)
    loss.backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_launch = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'{steps} steps: host launch path {t_launch / steps * 1e3:.3f} ms/step, with the GPU drained {t_all / steps * 1e3:.3f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=30, max_name_column_width=70))
