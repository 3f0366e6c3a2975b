"""Lift + splat at config-4 shapes: materialised lift followed by mf_bev_splat_* vs the fused mf_bev_lift_splat_*."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoforce_amd import _timing, splat, synthetic as syn
from monoforce_amd.terrain_encoder import LiftSplatShoot
DEV = 'cuda'
gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
enc = LiftSplatShoot(gc, dict(final_dim=(256, 512))).to(DEV)
for B in (1, 8):
    calib = [t.to(DEV) for t in syn.lss_camera_rig(B)]
    with torch.no_grad():
        geom = enc.get_geometry(*calib)
    _, N, D, fH, fW, _ = geom.shape
    C = 64
    plan = splat.SplatPlan(geom, enc.dx, enc.bx, enc.nx)
    depth = torch.rand(B * N, D, fH, fW, device=DEV).softmax(dim=1).requires_grad_(True)
    ctx = torch.randn(B * N, C, fH, fW, device=DEV, requires_grad=True)
    w = torch.randn(B, C, 256, 256, device=DEV)
    def unfused():
        x = (depth.unsqueeze(1) * ctx.unsqueeze(2)).view(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2)
        return splat._Pool.apply(x, plan)
    def fused():
        return splat._LiftPool.apply(depth, ctx, plan)
    for name, fn in (('lift + splat', unfused), ('fused lift-splat', fused)):
        for _ in range(3):
            out = fn(); (out * w).sum().backward()
        torch.cuda.synchronize()
        # (a) what the caller sees: events around the Python calls with the GPU drained before each (host launch path included);
        # (b) the hand-written kernels alone: HIP events around the C-ABI launches, 10 iterations queued back to back
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(10):
            depth.grad = ctx.grad = None
            e[0].record(); out = fn(); e[1].record(); out.backward(w); e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        _timing.start()
        for _ in range(10):
            depth.grad = ctx.grad = None
            fn().backward(w)
        k = {n: sum(v) / len(v) * 1e3 for n, v in _timing.stop().items()}
        kern = '  kernels: ' + ', '.join(f'{n} {v:.1f} us' for n, v in k.items())
        print(f'B={B} {name:18s} forward {tf / 10 * 1e3:7.1f} us   backward {tb / 10 * 1e3:7.1f} us (call to call, drained GPU){kern}', flush=True)
    # the geometry side: get_geometry + key/CSR passes from the geometry tensor vs the plan straight from the camera models
    def plan_geom():
        return splat.SplatPlan(enc.get_geometry(*calib), enc.dx, enc.bx, enc.nx)
    def plan_cams():
        return enc.splat_plan(*calib)
    for name, fn in (('get_geometry + plan', plan_geom), ('plan from cameras', plan_cams)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f'B={B} {name:20s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us on the device, {(time.perf_counter() - t0) / 20 * 1e6:7.1f} us wall', flush=True)
    # the same plan rebuilt in ONE persistent workspace (LiftSplatShoot.cache_plan = False: what a train loop with per-sample augmentation
    # runs), launch by launch and as a hipGraph replay (the plan has no host synchronisation: inside a captured train step its five
    # launches are graph nodes) -- the replay is the device-side cost of the pass without the host's launch pacing
    enc.cache_plan = False
    for _ in range(3):
        enc.splat_plan(*calib)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(20):
        enc.splat_plan(*calib)
    e1.record(); torch.cuda.synchronize()
    print(f'B={B} plan, one workspace     {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us on the device, {(time.perf_counter() - t0) / 20 * 1e6:7.1f} us wall', flush=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        enc.splat_plan(*calib)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
        enc.splat_plan(*calib)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f'B={B} plan, hipGraph replay   {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us on the device per replay (clear + 4 kernels)', flush=True)
    enc.cache_plan = True
