"""bench.py's multi-rank path on the one-GPU test rig: ranks share GPU 0 and exchange through gloo
(MF_BENCH_SINGLE_DEVICE / MF_BENCH_BACKEND); on an 8-GPU node the same code runs one rank per GPU over RCCL."""
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK') + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=900)


def _last_line_and_detail(r):
    last = r.stdout.rstrip('\n').splitlines()[-1]              # the driver parses the LAST stdout line: compact, < 4 KB
    line = json.loads(last)
    assert len(last) < 4096
    return line, json.load(open(line['detail'] if os.path.isabs(line['detail']) else os.path.join(REPO, line['detail'])))


def test_bench_spawns_its_own_ranks():
    """`bench.py --gpus 2` with no launcher around it starts two ranks itself and says so in the JSON line."""
    with tempfile.TemporaryDirectory() as td:
        r = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '64', '--no-others', '--no-cpu-baseline',
                        '--detail', os.path.join(td, 'd.json')], {'MF_BENCH_SINGLE_DEVICE': '1', 'MF_BENCH_BACKEND': 'gloo'})
        assert r.returncode == 0, r.stderr[-2000:]
        out, full = _last_line_and_detail(r)
    assert out['n_gpus'] == 2 and out['world_size'] == 2 and out['backend'] == 'gloo'
    assert out['scaling'] == 'weak' and out['config']['rollouts_per_gpu'] == 64 and out['config']['rollouts_total'] == 128
    assert out['value'] > 0 and set(out['roofline']['per_kernel_ms']) >= {'rollout_fwd_kernel', 'rollout_bwd_kernel'}
    assert set(full['roofline']['per_kernel']) >= {'rollout_fwd_kernel', 'rollout_bwd_kernel'} and 'cpu_baseline' not in out


def test_bench_refuses_a_different_rank_count():
    r = _run_bench(['--gpus', '2', '--steps', '1', '--warmup', '0', '--no-others', '--no-cpu-baseline'], {'WORLD_SIZE': '1'})
    assert r.returncode == 2 and 'refusing' in r.stderr
    if torch.cuda.device_count() < 3:       # more ranks than GPUs, no test rig: refused, not silently shrunk
        r = _run_bench(['--gpus', '3', '--steps', '1', '--warmup', '0', '--no-others', '--no-cpu-baseline'],
                       drop=('MF_BENCH_SINGLE_DEVICE',))
        assert r.returncode == 2 and 'refusing' in r.stderr


def _fit_worker(rank, world, store, B, T, out_path):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from bench import build_problem
    from monoforce_amd import synthetic as syn
    from monoforce_amd.train import TerrainFitProblem
    if world > 1:
        dist.init_process_group('gloo', init_method='file://' + store, rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    # bench.py's arrangement: ONE terrain (terrain_seed 0) on every rank, controls seeded by the rank
    if world > 1:
        _, dp, _, _, z, mu, ctrl = build_problem(B, T, 4, dev, 1, seed=rank, terrain_seed=0)
    else:                                   # the single-process equivalent: both ranks' rollouts in one batch
        _, dp, _, _, z, mu, c0 = build_problem(B, T, 4, dev, 1, seed=0, terrain_seed=0)
        ctrl = torch.cat([c0, syn.const_controls(B, T, seed=1)])
    z_true = syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(dev)
    prob = TerrainFitProblem(dp, z_true, mu.to(dev), ctrl.to(dev))
    zl, ml = z.to(dev).clone().requires_grad_(True), mu.to(dev).clone().requires_grad_(True)
    loss = prob.step(zl, ml)
    if rank == 0:
        torch.save({'loss': loss.detach().cpu(), 'gz': zl.grad.cpu(), 'gmu': ml.grad.cpu(), 'fast': prob.fast_exchange}, out_path)
    if world > 1:
        dist.destroy_process_group()


def test_two_rank_terrain_gradient_equals_single_process():
    """The one exchange step of the backward: two ranks (sharing GPU 0, gloo) fit ONE terrain to their own rollouts; the
    rank-averaged loss and gradient equal the single-process run over both ranks' rollouts."""
    import torch.multiprocessing as mp
    B, T = 48, 100
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_fit_worker, args=(2, os.path.join(td, 'store'), B, T, os.path.join(td, 'two.pt')), nprocs=2, join=True)
        _fit_worker(0, 1, None, B, T, os.path.join(td, 'one.pt'))
        two, one = torch.load(os.path.join(td, 'two.pt')), torch.load(os.path.join(td, 'one.pt'))
    assert two['fast'] is True            # gradients and loss went through ONE in-place collective (no pack / unpack copies)
    assert abs(float(two['loss']) - float(one['loss'])) <= 1e-5 * abs(float(one['loss']))
    for k in ('gz', 'gmu'):
        scale = float(one[k].abs().max())
        assert scale > 0 and float((two[k] - one[k]).abs().max()) <= 2e-4 * scale, k
