"""CPU tests: losses vs the reference's golden vectors; sharding + the gradient all-reduce on a 2-process gloo group."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import helpers as hp


def test_physics_loss_matches_reference():
    from monoforce_amd.losses import physics_loss
    g = hp.load('physics_loss')
    X = torch.as_tensor(g['X']).requires_grad_(True)
    loss = physics_loss([X], [torch.as_tensor(g['Xgt'])], torch.as_tensor(g['pred_ts']), torch.as_tensor(g['gt_ts']), gamma=0.9)
    loss.backward()
    assert abs(float(loss) - float(g['loss'])) <= 1e-6 * abs(float(g['loss']))
    assert hp.rel_err(X.grad, g['g_X']) <= 1e-6


def test_hm_loss_and_tv():
    from monoforce_amd.losses import hm_loss, total_variation
    a = torch.tensor([[[1.0, 2.0], [float('nan'), 4.0]]]); b = torch.tensor([[[1.5, 2.0], [3.0, float('nan')]]])
    assert abs(float(hm_loss(a, b)) - 0.125) < 1e-7                      # only the two finite pairs count
    w = torch.tensor([[[2.0, 1.0], [1.0, 1.0]]])
    assert abs(float(hm_loss(a, b, w)) - 0.5) < 1e-7
    assert abs(float(total_variation(torch.tensor([[0.0, 1.0], [3.0, 1.0]]))) - (1 + 2 + 3 + 0) / 4) < 1e-7


def test_shard_range_covers_everything():
    from monoforce_amd.dist import shard_range
    for n in (1, 7, 8, 1024, 8192, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from monoforce_amd import dist as mfd
    from oracle import dphysics_oracle as orc
    from monoforce_amd import synthetic as syn
    mfd.init(backend='gloo')
    assert mfd.world() == world and mfd.rank() == rank
    # the N>1 data path: shard B rollouts, compute local terrain gradients (CPU oracle stands in for the kernel here),
    # sum them with the flat-bucket all-reduce; result must equal the single-process gradient over all rollouts.
    pts, masks = syn.robot_points_4()
    B, T, dt = 6, 25, torch.float64
    z = (syn.bump_terrain(syn.bump_params(3), 1.6, 0.1, dt) * 0.3).requires_grad_(True)
    mu = syn.wave_friction(1.6, 0.1, dtype=dt).requires_grad_(True)
    ctrl = mfd.shard(syn.const_controls(B, T, seed=1, dtype=dt))
    spec = hp.spec_from(pts, masks, 1, 0.1, 1.6)
    b = ctrl.shape[0]
    (Xs, _, _, _), _ = orc.rollout(spec, z.unsqueeze(0).expand(b, -1, -1), ctrl, friction=mu.unsqueeze(0).expand(b, -1, -1))
    (Xs ** 2).sum().backward()
    bucket = mfd.allreduce_sum_([z.grad, mu.grad])
    bucket = mfd.allreduce_sum_([z.grad * 0 + 1.0, None], None)     # None entries are skipped
    if rank == 0:
        np.savez(out, gz=z.grad.numpy(), gmu=mu.grad.numpy())
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce(tmp_path):
    from monoforce_amd import synthetic as syn
    from oracle import dphysics_oracle as orc
    out = str(tmp_path / 'g.npz')
    port = 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    pts, masks = syn.robot_points_4()
    B, T, dt = 6, 25, torch.float64
    z = (syn.bump_terrain(syn.bump_params(3), 1.6, 0.1, dt) * 0.3).requires_grad_(True)
    mu = syn.wave_friction(1.6, 0.1, dtype=dt).requires_grad_(True)
    ctrl = syn.const_controls(B, T, seed=1, dtype=dt)
    (Xs, _, _, _), _ = orc.rollout(hp.spec_from(pts, masks, 1, 0.1, 1.6), z.unsqueeze(0).expand(B, -1, -1), ctrl,
                                   friction=mu.unsqueeze(0).expand(B, -1, -1))
    (Xs ** 2).sum().backward()
    assert hp.rel_err(got['gz'], z.grad) <= 1e-10
    assert hp.rel_err(got['gmu'], mu.grad) <= 1e-10


def _ddp_worker(rank, world, port, out, defer=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from monoforce_amd import dist as mfd
    mfd.init(backend='gloo')
    torch.manual_seed(0)                                   # identical initial weights on every rank
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
    unused = torch.nn.Parameter(torch.ones(3))             # a parameter no loss touches: exchanged as zeros
    params = list(net.parameters()) + [unused]
    gb = mfd.GradBuckets(params, bucket_mb=0.0001)          # ~100 bytes per bucket: several buckets, launched during backward
    assert len(gb.buckets) >= 3
    gb.defer = defer                                       # True: the hooks only count; pack() packs, exchange() runs the collectives
    opt = torch.optim.SGD(params, lr=0.1)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(12, 8, generator=g), torch.randn(12, 1, generator=g)
    lo, hi = mfd.shard_range(12)
    for _ in range(3):
        gb.zero()
        ((net(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
        if defer:                                          # the order of a step replayed as two hipGraphs around its exchange
            assert all(b['work'] is None and not b['launched'] for b in gb.buckets)
            gb.pack()
            gb.exchange()
        gb.finish()
        opt.step()
    if rank == 0:
        np.savez(out, **{f'p{i}': p.detach().numpy() for i, p in enumerate(params)}, g0=params[0].grad.numpy())
    torch.distributed.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize('defer', [False, True])
def test_two_rank_gloo_bucketed_overlapped_gradient_exchange(tmp_path, defer):
    """GradBuckets (gradients as views of flat buckets, all-reduce launched from autograd hooks -- or, deferred, after the
    backward: the graph-replayed step's order --, averaged) == single-process training on the whole batch."""
    out = str(tmp_path / 'ddp.npz')
    port = 31000 + os.getpid() % 2000 + (2000 if defer else 0)
    mp.spawn(_ddp_worker, args=(2, port, out, defer), nprocs=2, join=True)
    got = np.load(out)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
    unused = torch.nn.Parameter(torch.ones(3))
    params = list(net.parameters()) + [unused]
    opt = torch.optim.SGD(list(net.parameters()), lr=0.1)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(12, 8, generator=g), torch.randn(12, 1, generator=g)
    for _ in range(3):
        opt.zero_grad()
        # mean over the two equally sized shards of per-shard means == mean over the batch
        ((net(X) - Y) ** 2).mean().backward()
        last = params[0].grad.clone()
        opt.step()
    for i, p in enumerate(params):
        assert np.allclose(got[f'p{i}'], p.detach().numpy(), rtol=1e-5, atol=1e-6), i
    assert np.allclose(got['g0'], last.numpy(), rtol=1e-5, atol=1e-7)
