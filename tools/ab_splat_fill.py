"""What bounds the fused lift-splat forward at config-4 shapes: the dense output's store pattern or the point work?
Times, per batch size: (a) a plain zero fill of the output tensor (torch's fill kernel: the store bandwidth the box gives a
contiguous 16-byte-per-lane fill), (b) the fused forward on the camera rig's plan, (c) the fused forward on a plan whose points were
all dropped (geometry shifted off the grid: every tile takes the empty-tile path = the kernel's own fill), and the fraction of
64-voxel tiles that hold points.  AB_B=1,8"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoforce_amd import _timing, splat, synthetic as syn
from monoforce_amd.terrain_encoder import LiftSplatShoot
DEV = 'cuda'
gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
enc = LiftSplatShoot(gc, dict(final_dim=(256, 512))).to(DEV)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
    return ts[n // 2], ts[0]


for B in [int(v) for v in os.environ.get('AB_B', '1,8').split(',')]:
    calib = [t.to(DEV) for t in syn.lss_camera_rig(B)]
    with torch.no_grad():
        geom = enc.get_geometry(*calib)
    _, N, D, fH, fW, _ = geom.shape
    C = 64
    plan = splat.SplatPlan(geom, enc.dx, enc.bx, enc.nx)
    plan_empty = splat.SplatPlan(geom + 1000.0, enc.dx, enc.bx, enc.nx)
    keys = plan.keys()
    kept = keys[keys >= 0].long()
    tiles = torch.unique(kept // 64).numel()
    vox = torch.unique(kept).numel()
    n_tiles = B * plan.nz * plan.nx * plan.ny // 64
    depth = torch.rand(B * N, D, fH, fW, device=DEV).softmax(dim=1)
    ctx = torch.randn(B * N, C, fH, fW, device=DEV)
    out = torch.empty(B, C, 256, 256, device=DEV)
    mb = out.numel() * 4 / 1e6
    with torch.no_grad():
        for name, fn in (('zero fill of the output (torch)', lambda: out.zero_()),
                         ('fused forward, rig plan', lambda: splat._LiftPool.apply(depth, ctx, plan)),
                         ('fused forward, all points dropped', lambda: splat._LiftPool.apply(depth, ctx, plan_empty))):
            med, best = timed(fn)
            if 'fused' in name:      # the kernel alone: HIP events around the C-ABI launch
                _timing.start()
                for _ in range(20):
                    fn()
                ks = sorted(_timing.stop()['lift_splat_fwd_kernel'])
                med, best = ks[len(ks) // 2] * 1e3, ks[0] * 1e3
            print(f'B={B} {name:36s} median {med:7.1f} us  min {best:7.1f} us  ({mb / med:6.2f} TB/s of the {mb:.0f} MB output at the median)', flush=True)
    print(f'B={B} kept points {kept.numel()} of {keys.numel()}, occupied voxels {vox}, occupied tiles {tiles} of {n_tiles} ({tiles / n_tiles:.1%})', flush=True)
