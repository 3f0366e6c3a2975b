#!/usr/bin/env python3
"""Secondary measurements (not the bench.py headline): per-kernel time / algorithmic GB/s / HBM-roofline fraction for
  * the BEV splat at the BASELINE config-4 shapes (4 cams x 256x512 /16, D=59, C=64, 256x256 BEV): prepare, fwd, bwd
  * the rollout forward/backward over a batch sweep, N sweep, both integrators, shared vs per-rollout maps.
Prints one JSON object per line.  Usage: python tools/bench_kernels.py [splat] [rollout] [--reps 20]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

PEAK = 8000.0
DEV = 'cuda'


def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def splat(reps, B=1):
    from monoforce_amd import synthetic as syn
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    from monoforce_amd.splat import SplatPlan, voxel_pooling
    gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
    m = LiftSplatShoot(gc, dict(final_dim=(256, 512)), build_backbones=False).to(DEV)
    rig = [t.to(DEV) for t in syn.lss_camera_rig(B, 4, 256, 512, 300.0)]
    geom = m.get_geometry(*rig)
    C = 64
    x = torch.randn(*geom.shape[:-1], C, device=DEV)
    P = x.numel() // C
    out_elems = B * C * 256 * 256
    plan = SplatPlan(geom, m.dx, m.bx, m.nx)
    t_prep = timeit(lambda: SplatPlan(geom, m.dx, m.bx, m.nx), reps)
    t_fwd = timeit(lambda: voxel_pooling(None, x, None, None, None, plan=plan), reps)
    xg = x.clone().requires_grad_(True)
    out = voxel_pooling(None, xg, None, None, None, plan=plan)
    gout = torch.randn_like(out)
    t_bwd = timeit(lambda: torch.autograd.grad(out, xg, gout, retain_graph=True), reps)
    fwd_bytes = 4 * C * P + 12 * P + 4 * out_elems          # read x + geom, write the dense grid
    bwd_bytes = 4 * out_elems + 4 * C * P
    for name, t, by in (('splat_prepare', t_prep, 12 * P + 8 * P), ('splat_fwd', t_fwd, fwd_bytes), ('splat_bwd', t_bwd, bwd_bytes)):
        print(json.dumps({'kernel': name, 'B': B, 'points': P, 'C': C, 'ms': t, 'algorithmic_MB': by / 1e6,
                          'GB/s': by / t / 1e6, 'frac_hbm': by / t / 1e6 / PEAK}))


def rollout(reps):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_problem, fwd_bytes_per_rollout_step, bwd_bytes_per_rollout_step
    from monoforce_amd import _timing
    T = 500
    cases = [(B, 4, 1, True) for B in (256, 1024, 4096, 16384, 65536)] + [(1024, 4, 0, True), (1024, 4, 1, False),
             (1024, 32, 1, True), (1024, 175, 1, True), (1024, 223, 1, True), (16384, 4, 0, True)]
    for B, N, integ, shared in cases:
        cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, T, N, DEV, integ)
        zl, ml = z.to(DEV).clone().requires_grad_(True), mu.to(DEV).clone().requires_grad_(True)
        cd = ctrl.to(DEV)

        def step():
            if shared:
                zi, mi = zl.unsqueeze(0), ml.unsqueeze(0)
            else:
                zi, mi = zl.unsqueeze(0).repeat(B, 1, 1), ml.unsqueeze(0).repeat(B, 1, 1)
            (Xs, _, _, _), _ = dp(zi, cd, friction=mi)
            (Xs[:, ::10] ** 2).mean().backward()
        step(); step()
        _timing.start()
        for _ in range(max(reps // 4, 3)):
            step()
        k = {n: float(np.mean(v)) for n, v in _timing.stop().items()}
        f, b = k['rollout_fwd_kernel'], k['rollout_bwd_kernel']
        fb, bb = fwd_bytes_per_rollout_step(N) * B * T, bwd_bytes_per_rollout_step(N) * B * T
        print(json.dumps({'kernel': 'rollout', 'B': B, 'T': T, 'N': N, 'integrator': integ, 'shared_map': shared,
                          'fwd_ms': f, 'fwd_Msteps_s': B * T / f / 1e3, 'fwd_GB/s': fb / f / 1e6, 'fwd_frac': fb / f / 1e6 / PEAK,
                          'bwd_ms': b, 'bwd_GB/s': bb / b / 1e6, 'bwd_frac': bb / b / 1e6 / PEAK}))
        del dp, zl, ml, cd


if __name__ == '__main__':
    reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 20
    what = [a for a in sys.argv[1:] if a in ('splat', 'rollout')] or ['splat', 'rollout']
    if 'splat' in what:
        splat(reps, 1); splat(reps, 8)
    if 'rollout' in what:
        rollout(reps)
