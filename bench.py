#!/usr/bin/env python3
"""bench.py -- rollout-steps/s of the DPhysics hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic input: B rollouts x T Euler steps x N contact points on
a 256x256 terrain.  `value` = (B * T * world_size * K) / wall time of the K timed steps (barrier + synchronize on both
sides, max over ranks), inputs resident in HBM, outputs allocated inside the timed region (the API returns fresh
tensors).  Rollouts are independent, so ranks shard the batch with no data-path collective (weak scaling: B rollouts per
GPU); the backward workloads all-reduce the gradient of what the ranks share (terrain grids / encoder weights) over RCCL.

Workloads (`--workload`, default c3f = the shape the metric is quoted on):
  c2   BASELINE configs[1]   256 rollouts x 500 steps, forward
  c3f  north_star shape      1024 rollouts x 500 steps x 4 contact points, forward
  c3   BASELINE configs[2]   same + physics-loss backward to the (shared) terrain and friction grids
  c4   BASELINE configs[3/4] TerrainEncoder (4 cams 3x256x512 -> 256x256 BEV) + 1024 rollouts, end-to-end train step

Extra objects on the JSON line:
  roofline      dominant hand-written kernel of the step: algorithmic bytes per launch (80 + 56 N bytes per
                rollout-step forward, 160 + 120 N backward) / its average launch duration, measured live with HIP events
                on the launch stream (monoforce_amd/_timing.py), against the 8 TB/s HBM peak.  `traffic` = HBM bytes
                per launch from rocprofv3 PMC passes of this same command (profiles/hbm_traffic.json; null if absent).
  cpu_baseline  the CPU oracle (oracle/dphysics_oracle.py, a torch-CPU port of the reference algorithm) timed on this
                box's host cores on a bounded sample of the same workload.  Reported, not the target.
  other_workloads  (default run, 1 GPU only) short runs of c2 and c3, same accounting, and `shoot`: trajectory shooting of
                16384 sampled control sequences on the same terrain through the kernel's path-cost mode (rollout + path
                costs + argmin, end to end; monoforce_amd/planner.py).
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
    """upstream grads 72 + 24 N, controls 8, saved state 72, re-gather 32 N, scatter RMW 64 N, grad-controls 8."""
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
            orc.rollout(spec, zb, ctrl, friction=mb)
            times.append(time.perf_counter() - t0)
    best = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return dict(value=Bs * T / best, unit='rollout-steps/s', cores=torch.get_num_threads(), kind='port',
                sample=f'oracle/dphysics_oracle.py (torch-CPU port), B={Bs} x T={T} x N={N}, 256x256 shared map, forward '
                       f'no_grad, median of {max(len(times) - 1, 1)} runs after 1 warm-up; os.cpu_count()={os.cpu_count()}')


class Runner:
    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback)'
        if os.environ.get('MF_BENCH_SINGLE_DEVICE'):      # test rig: all ranks on GPU 0, gloo instead of RCCL
            local_rank = 0
        torch.cuda.set_device(local_rank)
        self.dev = torch.device('cuda', local_rank)
        self.backend = os.environ.get('MF_BENCH_BACKEND', 'nccl')      # "nccl" is RCCL on ROCm
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if self.backend == 'nccl':
                dist.init_process_group('nccl', device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
        assert self.world == args.gpus or self.world == 1, f'--gpus {args.gpus} but WORLD_SIZE={self.world}'
        traffic_file = os.path.join(REPO, 'profiles', 'hbm_traffic.json')
        self.traffic = json.load(open(traffic_file)) if os.path.exists(traffic_file) else {}

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def run(self, name, steps, warmup, with_sweep=False):
        from monoforce_amd import _timing
        args, dev, world, rank = self.args, self.dev, self.world, self.rank
        wl = dict(WORKLOADS[name])
        if args.batch:
            wl['B'] = args.batch
        if args.points:
            wl['N'] = args.points
        B, T, N = wl['B'], wl['T'], wl['N']
        cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, T, N, dev, args.integrator, seed=rank)
        dp.block = args.block
        zd = z.to(dev).unsqueeze(0)       # ONE terrain shared by the rollouts ([1,H,W] map + [B,T,2] controls)
        md = mu.to(dev).unsqueeze(0)
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

        for _ in range(warmup):
            step()
        self.barrier()
        _timing.start()             # HIP events around every C-ABI launch, on the stream the kernel is launched on
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.barrier()
        elapsed = time.perf_counter() - t0
        kern = {k: float(np.mean(v)) for k, v in _timing.stop().items()}       # average launch duration per kernel, ms
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([elapsed], device=dev if self.backend == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())

        P4 = 4 * 59 * 16 * 32                                  # frustum points per sample at the config-4 shapes
        alg = {'rollout_fwd_kernel': fwd_bytes_per_rollout_step(N) * B * T, 'rollout_bwd_kernel': bwd_bytes_per_rollout_step(N) * B * T,
               'splat_fwd_kernel': 4 * 64 * P4 + 12 * P4 + 4 * 64 * 256 * 256, 'splat_bwd_kernel': 4 * 64 * 256 * 256 + 4 * 64 * P4,
               'splat_prepare': 20 * P4,
               # lift fused into the splat: depth 4 P + context 4*64*pixels + out / the reverse + the voxel-major rows
               'lift_splat_fwd_kernel': 4 * P4 + 4 * 64 * (P4 // 59) + 4 * 64 * 256 * 256,
               'lift_splat_bwd_kernel': 4 * 64 * 256 * 256 + 8 * P4 + 2 * 4 * 64 * (P4 // 59),
               'physics_loss_fwd': 12 * 50 * B * 2, 'physics_loss_bwd': 12 * 50 * B * 2}
        kern = {k: v for k, v in kern.items() if k in alg}
        dom = max(kern, key=kern.get)                          # the dominant hand-written kernel of the step
        per_kernel = {k: {'ms': v, 'algorithmic_bytes': alg[k], 'GB/s': alg[k] / (v * 1e-3) / 1e9,
                          'frac': alg[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS} for k, v in kern.items()}
        achieved = per_kernel[dom]['GB/s']
        mode = 'encoder train step (fwd+bwd+Adam)' if wl.get('encoder') else 'forward+backward' if wl['backward'] else 'forward'
        integ = 'odeint-euler (reference default)' if args.integrator == 1 else 'dynamics()'
        res = {
            'value': B * T * world * steps / elapsed, 'steps': steps, 'warmup': warmup, 'ms_per_step': elapsed / steps * 1e3,
            'config': {'workload': f'{name}: B={B}/GPU x T={T} x N={N} contact points, 256x256 grid (res 0.05 m), one shared '
                                   f'terrain+friction map, integrator={integ}, {mode}; {wl["desc"]}',
                       'rollouts_per_gpu': B, 'horizon': T, 'contact_points': N, 'grid': [256, 256],
                       'parallelism': f'rollout-sharded x{world}'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': self.traffic.get(name, {}).get(dom), 'kernel': dom, 'kernel_ms': kern[dom],
                         'algorithmic_bytes_per_launch': alg[dom], 'bytes_per_rollout_step': alg[dom] // (B * T),
                         'per_kernel': per_kernel},
        }
        if with_sweep and not wl['backward']:
            sweep = {}
            for Bs in (256, 1024, 4096, 8192, 16384, 32768, 65536):
                _, dps, _, _, _, _, cs = build_problem(Bs, T, N, dev, args.integrator, seed=0)
                cs = cs.to(dev)
                with torch.no_grad():
                    dps(zd, cs, friction=md); dps(zd, cs, friction=md)
                    _timing.start()
                    for _ in range(3):
                        dps(zd, cs, friction=md)
                    ms_ = float(np.mean(_timing.stop()['rollout_fwd_kernel']))
                gbs = fwd_bytes_per_rollout_step(N) * Bs * T / (ms_ * 1e-3) / 1e9
                sweep[str(Bs)] = {'kernel_ms': ms_, 'rollout_steps_per_s': Bs * T / (ms_ * 1e-3), 'GB/s': gbs, 'frac': gbs / HBM_PEAK_GBS}
                del dps, cs
            res['batch_sweep'] = sweep
            torch.cuda.empty_cache()      # return the sweep's multi-GB blocks now, not inside a later workload's timed region
            torch.cuda.synchronize(dev)
        return res, (N, T)


def shoot_workload(r, T, N, integ, B=16384, iters=8):
    """Sampling-based planning (monoforce_node.py:41-126): B control samples, one shared map, force path cost, argmin."""
    from monoforce_amd import _timing
    from monoforce_amd.planner import TrajectoryShooter, sample_controls
    dev = r.dev
    cfg, dp, _, _, z, mu, _ = build_problem(B, T, N, dev, integ, seed=0)
    zd, md = z.to(dev), mu.to(dev)
    c = sample_controls(B, cfg, dev, torch.Generator(device=dev).manual_seed(0))
    sh = TrajectoryShooter(dp, n_trajs=B, cost='force')
    for _ in range(3):
        sh.shoot(zd, friction=md, controls=c)
    torch.cuda.synchronize(dev)
    _timing.start()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = sh.shoot(zd, friction=md, controls=c)        # int(argmin) inside synchronises every iteration, like the node
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / iters * 1e3
    kms = float(np.mean(_timing.stop()['rollout_fwd_kernel']))
    return {'value': B * T / (ms * 1e-3), 'unit': 'rollout-steps/s', 'ms_per_step': ms,
            'workload': f'shoot: {B} sampled control sequences x T={T} x N={N}, one shared map, path-cost kernel + force cost + argmin',
            'per_kernel': {'rollout_fwd_kernel': {'ms': kms, 'bytes_per_rollout_step': 8 + 16 + 32 * N,
                                                  'GB/s': (8 + 16 + 32 * N) * B * T / (kms * 1e-3) / 1e9}}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c3f', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='override rollouts per GPU')
    ap.add_argument('--points', type=int, default=0, help='override contact points')
    ap.add_argument('--integrator', type=int, default=1, help='1 = odeint-euler (reference default), 0 = dynamics()')
    ap.add_argument('--block', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-others', action='store_true', help='skip the short c2 / c3 side runs of the default command')
    ap.add_argument('--sweep', action='store_true', help='also report a forward batch sweep (not part of the timed value)')
    args = ap.parse_args()

    r = Runner(args)
    res, (N, T) = r.run(args.workload, args.steps, args.warmup, with_sweep=args.sweep)
    others = {}
    if args.workload == 'c3f' and r.world == 1 and not args.no_others and not args.batch and not args.points:
        for name in ('c2', 'c3'):
            o, _ = r.run(name, max(args.steps // 3, 5), 3)
            others[name] = {'value': o['value'], 'unit': 'rollout-steps/s', 'ms_per_step': o['ms_per_step'],
                            'workload': o['config']['workload'], 'per_kernel': o['roofline']['per_kernel']}
        others['shoot'] = shoot_workload(r, T, N, args.integrator)
    if r.rank == 0:
        out = {'metric': 'rollout-steps/sec (batch x horizon) on 256x256 terrain', 'value': res['value'], 'unit': 'rollout-steps/s',
               'n_gpus': r.world, 'steps': res['steps'], 'warmup': res['warmup'], 'ms_per_step': res['ms_per_step'],
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': res['config'], 'roofline': res['roofline']}
        if 'batch_sweep' in res:
            out['batch_sweep'] = res['batch_sweep']
        if others:
            out['other_workloads'] = others
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(N, args.integrator, T) if r.world == 1 else None      # timed at N=1 only
        print(json.dumps(out))
    if r.world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
