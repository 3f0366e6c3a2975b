"""`torch.ops.monoforce.*` (monoforce_amd/ops.py): the registered ops give the results and gradients of the `DPhysics` /
`voxel_pooling` paths they wrap, and a forward + backward step through them can be captured into a hipGraph."""
import subprocess
import sys
import os

import pytest
import torch

from tests import helpers as hp
from tests.golden_state import given_state
from tests.test_rollout_gpu import make_dphysics

pytestmark = pytest.mark.gpu
DEV = 'cuda'
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('tag', ['f32', 'f64'])
@pytest.mark.parametrize('shared', [False, True])
def test_rollout_op_matches_dphysics(integ, tag, shared):
    from monoforce_amd import ops, synthetic as syn
    dt = hp.DT[tag]
    pts, masks = syn.robot_points_4()
    B, T = 6, 50
    nz = 1 if shared else B
    z = torch.stack([syn.bump_terrain(syn.bump_params(30 + k), 1.6, 0.1, torch.float64) * 0.3 for k in range(nz)]).to(dt)
    mu = torch.stack([syn.wave_friction(1.6, 0.1, 0.5, 1.0, 1.1 + 0.1 * k, 0.9, torch.float64) for k in range(nz)]).to(dt)
    ctrl = syn.varying_controls(B, T, seed=2, dtype=torch.float64).to(dt)
    dp = make_dphysics(pts, masks, integ, 0.1, 1.6)
    dp.dphys_cfg.traj_sim_time = T * dp.dphys_cfg.dt
    dp = type(dp)(dp.dphys_cfg, device=DEV)
    res = {}
    for how in ('module', 'op'):
        zd, md, cd = (t.to(DEV).requires_grad_(True) for t in (z, mu, ctrl))
        st = tuple(s.to(dt).to(DEV) for s in given_state(B))
        states, forces = (dp(zd, cd, state=st, friction=md) if how == 'module' else ops.rollout(dp, zd, cd, st, friction=md))
        outs = list(states) + list(forces)
        hp.probe_loss(outs, dt).backward()
        res[how] = [o.detach() for o in outs] + [zd.grad, md.grad, cd.grad, st[0]]
    tol = 1e-12 if tag == 'f64' else 1e-6
    for a, b in zip(res['op'], res['module']):
        assert a.shape == b.shape and hp.rel_err(a, b) <= tol, hp.rel_err(a, b)


def test_splat_op_matches_voxel_pooling():
    from monoforce_amd import ops
    from monoforce_amd.lss_utils import gen_dx_bx
    from monoforce_amd.splat import voxel_pooling
    dx, bx, nx = gen_dx_bx([-3.2, 3.2, 0.1], [-3.2, 3.2, 0.1], [-2.0, 2.0, 4.0])
    g = torch.Generator().manual_seed(0)
    geom = (torch.rand(2, 3, 10, 4, 6, 3, generator=g) * 7 - 3.5).to(DEV)
    x = torch.randn(2, 3, 10, 4, 6, 16, generator=g).to(DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    a = ops.splat(geom, xa, dx, bx, nx)
    b = voxel_pooling(geom, xb, dx.to(DEV), bx.to(DEV), nx.to(DEV))
    assert torch.equal(a, b)
    w = torch.randn_like(a)
    (a * w).sum().backward(); (b * w).sum().backward()
    assert torch.equal(xa.grad, xb.grad)


_CAPTURE = r'''
import sys, torch
sys.path.insert(0, %r)
from monoforce_amd import ops, synthetic as syn
from tests.test_rollout_gpu import make_dphysics
pts, masks = syn.robot_points_4()
B, T = 256, 100
dp = make_dphysics(pts, masks, 1, 0.05, 6.4)
dp.dphys_cfg.traj_sim_time = 1.0
dp = type(dp)(dp.dphys_cfg, device='cuda')
z = syn.bump_terrain(syn.bump_params(3), 6.4, 0.05).cuda().unsqueeze(0).requires_grad_(True)
mu = syn.wave_friction(6.4, 0.05).cuda().unsqueeze(0).requires_grad_(True)
ctrl = syn.const_controls(B, T, seed=1).cuda()
x0 = torch.zeros(B, 3, device='cuda'); R0 = torch.eye(3, device='cuda').repeat(B, 1, 1)
def step():
    z.grad = None; mu.grad = None
    (Xs, _, _, _), _ = ops.rollout(dp, z, ctrl, (x0.clone(), torch.zeros_like(x0), R0, torch.zeros_like(x0)), friction=mu)
    Xs[:, ::10].square().mean().backward()
    return z.grad, mu.grad
step(); step(); torch.cuda.synchronize()
ref = [g.clone() for g in step()]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    gz, gm = step()
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
ok = all(float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) for a, b in zip((gz, gm), ref))
print('CAPTURE_OK' if ok else 'CAPTURE_MISMATCH')
'''


def test_train_step_through_the_ops_can_be_graph_captured():
    """Forward + loss + backward through the registered ops, captured once and replayed as one hipGraph (round 1 noted that a
    captured train step dumped core in hipStreamEndCapture).  Run in a child process: a crash there must not take pytest down."""
    r = subprocess.run([sys.executable, '-c', _CAPTURE % REPO], capture_output=True, text=True, timeout=600)
    assert 'CAPTURE_OK' in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])


_FIT_GRAPH = r'''
import sys, torch
sys.path.insert(0, %r)
from bench import build_problem
from monoforce_amd import synthetic as syn
from monoforce_amd.train import TerrainFitProblem
dev = torch.device('cuda', 0)
cfg, dp, pts, masks, z, mu, ctrl = build_problem(512, 200, 4, dev, 1, seed=0)
z_true = syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(dev)
prob = TerrainFitProblem(dp, z_true, mu.to(dev), ctrl.to(dev), graph=True)
zl, ml = z.to(dev).clone().requires_grad_(True), mu.to(dev).clone().requires_grad_(True)
res = []
for eager in (True, False, False, True, False):       # launch by launch, captured + replayed, replayed, eager again, replayed
    loss = prob.step(zl, ml, eager=eager)
    torch.cuda.synchronize()
    res.append((float(loss), zl.grad.clone(), ml.grad.clone()))
with torch.no_grad():                                   # a changed terrain is seen by the replay (same tensors, new values)
    zl.add_(0.01 * torch.sin(torch.arange(zl.numel(), device=dev, dtype=zl.dtype) * 0.01).view_as(zl))
a = prob.step(zl, ml, eager=True); ga = zl.grad.clone()
b = prob.step(zl, ml); gb = zl.grad.clone()
torch.cuda.synchronize()
rel = lambda u, v: float((u - v).abs().max()) / max(float(v.abs().max()), 1e-30)
ok = all(abs(l - res[0][0]) <= 1e-5 * abs(res[0][0]) and rel(gz, res[0][1]) <= 2e-4 and rel(gm, res[0][2]) <= 2e-4 for l, gz, gm in res[1:])
ok = ok and abs(float(a) - float(b)) <= 1e-5 * abs(float(a)) and rel(gb, ga) <= 2e-4 and abs(float(a) - res[0][0]) > 1e-7 * abs(res[0][0])
print('FIT_GRAPH_OK' if ok else 'FIT_GRAPH_MISMATCH', [r[0] for r in res], float(a), float(b))
'''


def test_terrain_fit_step_replayed_as_one_graph_equals_the_eager_step():
    """`TerrainFitProblem(graph=True)`: forward + fused loss + backward captured once and replayed as one hipGraph per step give
    the loss and gradients of the launch-by-launch step, steps of both kinds can be mixed, and a replay sees new terrain
    values.  (Child process: a crash inside a capture must not take pytest down.)"""
    r = subprocess.run([sys.executable, '-c', _FIT_GRAPH % REPO], capture_output=True, text=True, timeout=600)
    assert 'FIT_GRAPH_OK' in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-3000:])
