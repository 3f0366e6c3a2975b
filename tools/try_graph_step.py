"""Which route of a train step survives hipGraph capture?  Each variant runs in a child process (a crash must not stop the
others).  History: the variants through `DPhysics.forward` (module_*, full) used to segfault inside hipStreamEndCapture while the
ones through the registered ops (ops_*) captured.  Cause (found by clearing module attributes one by one before the capture):
the module kept `self.z_grid / self.friction` as the caller's tensors -- views of leaves that require grad, whose gradient
accumulators (created on the default stream by the eager warm-up steps) stayed alive with them; the backward captured on the
side stream then had to be synchronised with the default stream.  With detached attributes every variant captures; the tool
stays as the regression check and prints eager vs replayed step times."""
import subprocess, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BODY = r'''
import sys, time, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, %r)
from bench import build_problem
from monoforce_amd import synthetic as syn, ops
from monoforce_amd.train import TerrainFitProblem
from monoforce_amd.losses import physics_loss_fused, physics_loss
dev = torch.device('cuda', 0)
variant = %r
B, T = (256, 100) if variant.endswith('_small') else (1024, 500)
cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, T, 4, dev, 1, seed=0)
z_true = syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(dev)
prob = TerrainFitProblem(dp, z_true, mu.to(dev), ctrl.to(dev))
zl, ml = z.to(dev).clone().requires_grad_(True), mu.to(dev).clone().requires_grad_(True)
cd = ctrl.to(dev)
def step():
    zl.grad = None; ml.grad = None
    if variant == 'full':
        return prob.step(zl, ml)
    if variant.startswith('ops'):
        x0 = torch.zeros(B, 3, device=dev); R0 = torch.eye(3, device=dev).repeat(B, 1, 1)
        states, _ = ops.rollout(dp, zl.unsqueeze(0), cd, (x0, torch.zeros_like(x0), R0, torch.zeros_like(x0)), friction=ml.unsqueeze(0))
    elif variant.startswith('module_state'):
        x0 = torch.zeros(B, 3, device=dev); R0 = torch.eye(3, device=dev).repeat(B, 1, 1)
        states, _ = dp(zl.unsqueeze(0), cd, state=(x0, torch.zeros_like(x0), R0, torch.zeros_like(x0)), friction=ml.unsqueeze(0))
    else:
        states, _ = dp(zl.unsqueeze(0), cd, friction=ml.unsqueeze(0))
    if variant in ('module_plain', 'module_plain_small', 'ops_big', 'ops_small', 'module_state', 'module_state_small'):
        loss = states[0][:, ::10].square().mean()
    elif variant == 'module_fused_loss':
        loss = physics_loss_fused(states, prob.states_gt, prob.pred_ts, prob.gt_ts, nearest=prob.nearest)
    elif variant == 'module_torch_loss':
        loss = physics_loss(states, prob.states_gt, prob.pred_ts, prob.gt_ts, nearest=prob.nearest.long())
    loss.backward()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize(); print(variant, 'eager ms/step %%.3f' %% ((time.perf_counter() - t0) / 30 * 1e3), flush=True)
ref = zl.grad.clone()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    lg = step()
print(variant, 'captured', flush=True)
for _ in range(3): g.replay()
torch.cuda.synchronize()
print(variant, 'replay ok, grad err %%.2e' %% float((zl.grad - ref).abs().max() / ref.abs().max()), flush=True)
t0 = time.perf_counter()
for _ in range(30): g.replay()
torch.cuda.synchronize(); print(variant, 'graph ms/step %%.3f' %% ((time.perf_counter() - t0) / 30 * 1e3), flush=True)
'''
for v in sys.argv[1:] or ['ops_small', 'ops_big', 'module_plain_small', 'module_state_small', 'module_state']:
    r = subprocess.run([sys.executable, '-c', BODY % (REPO, v)], capture_output=True, text=True, timeout=600)
    print(f'== {v}: rc={r.returncode}')
    print(r.stdout.strip())
    err = [l for l in r.stderr.splitlines() if 'Warning' not in l and 'amdgpu.ids' not in l and l.strip()]
    print('\n'.join(err[-12:]))
