"""The RCCL branches of the multi-GPU step on a ONE-GPU box: a one-rank `nccl` process group with every
collective forced (`monoforce_amd.dist.FORCE` / MF_DIST_FORCE, bench.py MF_BENCH_FORCE_DIST) -- `init_process_group('nccl',
device_id=...)`, the in-place `all_reduce(AVG)` on the backward's gradient buffer, the hooked bucket exchange, a hipGraph replay
followed by a collective on the same stream, and `bench.py --gpus 1` through its N > 1 code.  With one rank the collectives
are identities, so every result must equal the plain single-process run: what is tested is that the code RUNS on RCCL.
Each case runs in a child process (a process group is process-global state)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update({'MF_DIST_FORCE': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0', 'PYTHONPATH': REPO})
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, '-c', PRELUDE + textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=timeout, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


PRELUDE = '''
import os, socket, sys, torch
import torch.distributed as dist
from monoforce_amd import dist as mfdist
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); os.environ['MASTER_PORT'] = str(s.getsockname()[1])
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
mfdist.init('nccl', device=dev, force=True)
assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1 and mfdist.active()
'''


def test_one_rank_rccl_collectives_are_identities():
    _child('''
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(2 * 256 * 256 + 1, device=dev, generator=g)
    ref = a.clone()
    mfdist.allreduce_mean_inplace_(a)                       # ReduceOp.AVG on RCCL, in place
    torch.cuda.synchronize()
    assert torch.equal(a, ref)
    ts = [torch.randn(256, 256, device=dev, generator=g), torch.randn(256, 256, device=dev, generator=g), torch.randn(1, device=dev, generator=g)]
    refs = [t.clone() for t in ts]
    bucket = mfdist.allreduce_sum_(ts, None, average=True)   # pack -> all_reduce(SUM) -> divide -> unpack
    torch.cuda.synchronize()
    assert bucket is not None and all(torch.equal(t, r) for t, r in zip(ts, refs))
    dist.destroy_process_group()
    print('ok')
    ''')


def test_one_rank_bucketed_exchange_equals_plain_backward():
    _child('''
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Linear(256, 8)).to(dev)
    x = torch.randn(32, 64, device=dev)
    net(x).square().mean().backward()
    plain = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    buckets = mfdist.GradBuckets(list(net.parameters()), bucket_mb=0.1, average=True)      # several buckets: hooks fire mid-backward
    assert len(buckets.buckets) >= 2
    for it in range(2):
        buckets.zero()
        net(x).square().mean().backward()                    # async all_reduce launched from the hooks (RCCL's own stream)
        assert any(b['work'] is not None for b in buckets.buckets)
        buckets.finish()
        torch.cuda.synchronize()
        for p, g in zip(net.parameters(), plain):
            assert torch.equal(p.grad, g)
    dist.destroy_process_group()
    print('ok')
    ''')


def test_graph_replay_followed_by_the_collective():
    """`TerrainFitProblem(graph=True)`: forward + loss + backward replayed as one hipGraph, then the in-place all_reduce(AVG) of
    the gradient buffer the graph wrote, on the same stream -- equal to the launch-by-launch step without a process group."""
    _child('''
    sys.path.insert(0, os.getcwd())
    from bench import build_problem
    from monoforce_amd import synthetic as syn
    from monoforce_amd.train import TerrainFitProblem
    B, T = 256, 120
    _, dp, _, _, z, mu, ctrl = build_problem(B, T, 4, dev, 1, seed=0)
    z_true = syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(dev)
    zl, ml = z.to(dev).clone().requires_grad_(True), mu.to(dev).clone().requires_grad_(True)
    mfdist.FORCE = False
    ref = TerrainFitProblem(dp, z_true, mu.to(dev), ctrl.to(dev))
    l0 = float(ref.step(zl, ml)); g0 = (zl.grad.clone(), ml.grad.clone())
    mfdist.FORCE = True
    prob = TerrainFitProblem(dp, z_true, mu.to(dev), ctrl.to(dev), graph=True)
    for it in range(4):
        loss = prob.step(zl, ml)                             # capture at the first call, replays afterwards
        torch.cuda.synchronize()
        assert prob.graph and prob.fast_exchange is True
        assert abs(float(loss) - l0) <= 1e-6 * abs(l0)
        for a, b in zip((zl.grad, ml.grad), g0):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    loss = prob.step(zl, ml, eager=True)                     # and launch by launch through the same exchange
    assert abs(float(loss) - l0) <= 1e-6 * abs(l0)
    dist.destroy_process_group()
    print('ok')
    ''')


def test_bench_through_its_multi_rank_code_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update({'MF_BENCH_FORCE_DIST': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '2', '--batch', '256',
                        '--no-others', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.rstrip('\n').splitlines()[-1]
    assert last.startswith('{') and len(last) < 4096, r.stdout[-600:]
    out = json.loads(last)
    assert out['backend'] == 'rccl' and out['world_size'] == 1 and out['n_gpus'] == 1
    assert out['comm_ms'] is not None and 0 < out['comm_ms'] < float('inf')      # (positive and finite; no timing bar: the host of a box can be slow)
    assert out['ms_per_step_ranks']['min'] <= out['ms_per_step_ranks']['max'] and out['value'] > 0
