"""Cycle accounting of the streaming backward (A/B build with -DMF_STREAM_PROFILE: tools/build_variant.sh prof "-DMF_STREAM_PROFILE"
rollout_bwd_cp_stream_fast.hip; run with MONOFORCE_HIP_LIB=gpurun_in_ab/prof/libmonoforce_hip.so): per launch and workgroup, the
cycles the fetching waves spend waiting for room in the ring / for their turn to publish / in total, and the cycles the computing wave
waits for steps / runs in total.  AB_B=256,1024"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _lib, _timing
L = _lib.lib()
for B in [int(x) for x in os.environ.get('AB_B', '256,1024,2048').split(',')]:
    cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, 4, 'cuda', 1)
    dp.return_forces = False
    zl, ml = z.cuda().clone().requires_grad_(True), mu.cuda().clone().requires_grad_(True)
    cd = ctrl.cuda()
    def step():
        (Xs, _, _, _), _ = dp(zl.unsqueeze(0), cd, friction=ml.unsqueeze(0))
        (Xs[:, ::10] ** 2).mean().backward()
    step(); step(); torch.cuda.synchronize()
    L.mf_debug_stream_profile(None, 1)
    n = 4
    _timing.start()
    for _ in range(n): step()
    k = {nm: float(np.mean(v)) for nm, v in _timing.stop().items()}
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    assert L.mf_debug_stream_profile(out, 0) == 0
    wg = (B + 3) // 4
    per = [v / (n * wg) for v in out]      # timer ticks per launch and workgroup
    print(f'B {B}: bwd {k["rollout_bwd_kernel"]:.4f} ms | fetcher0 room {per[0]:.0f} publish {per[1]:.0f} total {per[2]:.0f} | fetcher1 room {per[4]:.0f} '
          f'publish {per[5]:.0f} total {per[6]:.0f} | compute wait {per[8]:.0f} total {per[9]:.0f}  (ticks per launch and workgroup)', flush=True)
