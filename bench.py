#!/usr/bin/env python3
"""bench.py -- rollout-steps/s of the DPhysics hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic rollouts: B rollouts x T Euler steps x N contact
points on a 256x256 terrain.  `value` = (B * T * world_size * K) / wall time of the K timed steps, inputs resident in
HBM, outputs allocated inside the timed region (the API returns fresh tensors).  Rollouts are independent, so ranks
shard the batch with no data-path collective (weak scaling: B rollouts per GPU); the backward workload all-reduces
the shared-terrain gradient over RCCL.

Extra objects on the JSON line:
  roofline      algorithmic bytes per launch (80 + 56 N bytes per rollout-step forward) / average kernel
                duration measured with HIP events on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline  the CPU oracle (oracle/dphysics_oracle.py, a torch-CPU port of the reference algorithm) timed on
                this box's host cores on a bounded sample of the same workload.  Reported, not the target.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s HBM3E peak (~6.3 TB/s achievable)

WORKLOADS = {
    # name: (B per GPU, T, N, integrator, backward)
    'c2': dict(B=256, T=500, N=4, backward=False, desc='BASELINE configs[1]: 256 rollouts x 500 steps, forward'),
    'c3f': dict(B=1024, T=500, N=4, backward=False, desc='north_star shape: 1024 rollouts x 500 steps x 4 points, forward'),
    'c3': dict(B=1024, T=500, N=4, backward=True, desc='BASELINE configs[2]: 1024 rollouts x 500 steps, forward + backward to terrain'),
    'c4': dict(B=1024, T=500, N=4, backward=True, encoder=True,
               desc='BASELINE configs[3]/[4]: TerrainEncoder (4 cams 3x256x512 -> 256x256 BEV) + 1024 rollouts per GPU, end-to-end train step'),
}


def fwd_bytes_per_rollout_step(N):
    """controls 8 + states written 72 + forces written 24 N + map cells gathered 32 N."""
    return 80 + 56 * N


def bwd_bytes_per_rollout_step(N):
    return 160 + 120 * N


def build_problem(B, T, N, device, integ, seed=0):
    from monoforce_amd import synthetic as syn
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    if N == 4:
        pts, masks = syn.robot_points_4()
    else:
        pts, masks = syn.robot_points_box(N, seed=1, n_tracks=2)
    cfg = DPhysConfig(robot='tradr', grid_res=0.05, robot_points=pts, driving_parts=masks)
    cfg.use_odeint = (integ == 1)
    z = syn.bump_terrain(syn.bump_params(seed), 6.4, 0.05)
    mu = syn.wave_friction(6.4, 0.05)
    ctrl = syn.const_controls(B, T, seed=seed)
    dp = DPhysics(cfg, device=device) if device is not None else None
    return cfg, dp, pts, masks, z, mu, ctrl


def cpu_baseline(N, integ, T, budget_s=20.0):
    """Time the CPU oracle on a bounded sample (forward, no_grad), host threads as torch sees them."""
    from oracle import dphysics_oracle as orc      # checker / baseline only -- never on the product path
    Bs = 256
    cfg, _, pts, masks, z, mu, ctrl = build_problem(Bs, T, N, None, integ, seed=0)
    spec = orc.RolloutSpec(points=torch.as_tensor(pts), driving_parts=[torch.as_tensor(m) for m in masks],
                           robot_size_y=float(cfg.robot_size[1]), mass=cfg.robot_mass, grid_res=0.05, d_max=6.4,
                           integrator=integ)
    zb, mb = z.unsqueeze(0).expand(Bs, -1, -1), mu.unsqueeze(0).expand(Bs, -1, -1)
    times = []
    t_start = time.perf_counter()
    with torch.no_grad():
        while len(times) < 2 or (time.perf_counter() - t_start < budget_s and len(times) < 12):
            t0 = time.perf_counter()
            orc.rollout(spec, zb, ctrl)
            times.append(time.perf_counter() - t0)
    best = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return dict(value=Bs * T / best, unit='rollout-steps/s', cores=torch.get_num_threads(), kind='port',
                sample=f'oracle/dphysics_oracle.py (torch-CPU port), B={Bs} x T={T} x N={N}, 256x256 shared map, forward '
                       f'no_grad, median of {max(len(times) - 1, 1)} runs after 1 warm-up; os.cpu_count()={os.cpu_count()}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c3f', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='override rollouts per GPU')
    ap.add_argument('--points', type=int, default=0, help='override contact points')
    ap.add_argument('--integrator', type=int, default=1, help='1 = odeint-euler (reference default), 0 = dynamics()')
    ap.add_argument('--block', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sweep', action='store_true', help='also report a batch sweep (not part of the timed value)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback)'
    if os.environ.get('MF_BENCH_SINGLE_DEVICE'):      # test rig: all ranks on GPU 0, gloo instead of RCCL
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    backend = os.environ.get('MF_BENCH_BACKEND', 'nccl')      # "nccl" is RCCL on ROCm
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl['B'] = args.batch
    if args.points:
        wl['N'] = args.points
    B, T, N = wl['B'], wl['T'], wl['N']
    cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, T, N, dev, args.integrator, seed=rank)
    dp.block = args.block
    zd = z.to(dev).unsqueeze(0).expand(B, -1, -1)       # one terrain shared by the rollouts (stride-0 batch)
    md = mu.to(dev).unsqueeze(0).expand(B, -1, -1)
    cd = ctrl.to(dev)
    if wl.get('encoder'):
        from monoforce_amd.terrain_encoder import LiftSplatShoot
        from monoforce_amd.train import EncoderTrainStep, synthetic_encoder_batch
        torch.manual_seed(0)        # identical initial weights on every rank (DDP convention)
        gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
        enc = LiftSplatShoot(gc, dict(final_dim=(256, 512))).to(dev).train()
        ebatch = synthetic_encoder_batch(enc, dp, n_rollouts=B, device=dev, seed=rank)
        estep = EncoderTrainStep(enc, dp, lr=1e-4)
    elif wl['backward']:
        from monoforce_amd.train import TerrainFitProblem
        from monoforce_amd import synthetic as syn
        z_true = syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(dev)       # GT trajectories come from another terrain
        prob = TerrainFitProblem(dp, z_true, mu.to(dev), cd)
        zleaf = z.to(dev).clone().requires_grad_(True)
        mleaf = mu.to(dev).clone().requires_grad_(True)

    def step():
        if wl.get('encoder'):
            return estep.step(ebatch)
        if wl['backward']:
            return prob.step(zleaf, mleaf)
        with torch.no_grad():
            return dp(zd, cd, friction=md)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    from monoforce_amd import _timing
    _timing.start()             # HIP events around every C-ABI launch, on the stream the kernel is launched on
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kern = {k: float(np.mean(v)) for k, v in _timing.stop().items()}       # average launch duration per kernel, ms
    if world > 1:
        tt = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        units = B * T * world * args.steps
        alg = {'rollout_fwd_kernel': fwd_bytes_per_rollout_step(N) * B * T, 'rollout_bwd_kernel': bwd_bytes_per_rollout_step(N) * B * T}
        P4 = 4 * 59 * 16 * 32                                  # frustum points per sample at the config-4 shapes
        alg.update({'splat_fwd_kernel': 4 * 64 * P4 + 12 * P4 + 4 * 64 * 256 * 256, 'splat_bwd_kernel': 4 * 64 * 256 * 256 + 4 * 64 * P4,
                    'splat_prepare': 20 * P4})
        kern = {k: v for k, v in kern.items() if k in alg}
        dom = max(kern, key=kern.get)                      # the dominant hand-written kernel of the step
        kern_ms, alg_bytes = kern[dom], alg[dom]
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        per_kernel = {k: {'ms': v, 'algorithmic_bytes': alg[k], 'GB/s': alg[k] / (v * 1e-3) / 1e9,
                          'frac': alg[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS} for k, v in kern.items()}
        out = {
            'metric': 'rollout-steps/sec (batch x horizon) on 256x256 terrain',
            'value': units / elapsed, 'unit': 'rollout-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.workload}: B={B}/GPU x T={T} x N={N} contact points, 256x256 grid (res 0.05 m), '
                                   f'one shared terrain+friction map, integrator='
                                   f'{"odeint-euler (reference default)" if args.integrator == 1 else "dynamics()"}, '
                                   f'{"encoder train step (fwd+bwd+Adam)" if wl.get("encoder") else "forward+backward" if wl["backward"] else "forward"}; {wl["desc"]}',
                       'rollouts_per_gpu': B, 'horizon': T, 'contact_points': N, 'grid': [256, 256],
                       'parallelism': f'rollout-sharded x{world}'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                         'kernel': dom, 'kernel_ms': kern_ms, 'algorithmic_bytes_per_launch': alg_bytes,
                         'bytes_per_rollout_step': alg_bytes // (B * T), 'per_kernel': per_kernel},
        }
        if args.sweep and not wl['backward']:
            sweep = {}
            for Bs in (256, 1024, 4096, 16384, 65536):
                _, dps, _, _, _, _, cs = build_problem(Bs, T, N, dev, args.integrator, seed=0)
                zs, ms, cs = z.to(dev).unsqueeze(0).expand(Bs, -1, -1), mu.to(dev).unsqueeze(0).expand(Bs, -1, -1), cs.to(dev)
                with torch.no_grad():
                    dps(zs, cs, friction=ms); torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(3):
                        dps(zs, cs, friction=ms)
                    b.record(); torch.cuda.synchronize()
                ms_ = a.elapsed_time(b) / 3
                gbs = fwd_bytes_per_rollout_step(N) * Bs * T / (ms_ * 1e-3) / 1e9
                sweep[str(Bs)] = {'ms': ms_, 'rollout_steps_per_s': Bs * T / (ms_ * 1e-3), 'GB/s': gbs, 'frac': gbs / HBM_PEAK_GBS}
                del dps, zs, ms, cs
            out['batch_sweep'] = sweep
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(N, args.integrator, T)
        elif not args.no_cpu_baseline:
            out['cpu_baseline'] = None     # timed at N=1 only
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
