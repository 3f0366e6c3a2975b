"""The backward's analogue of tools/horizon_stats.py: control gradients per rollout over the FULL 500-step horizon, HIP float32 and the oracle's
own float32 autograd against the oracle's float64 autograd.    python tools/grad_horizon_stats.py [B] [H]"""
import os, sys
os.environ.setdefault('OMP_NUM_THREADS', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np, torch
torch.set_num_threads(min(int(os.environ['OMP_NUM_THREADS']), torch.get_num_threads()))
from tests.horizon_cases import grad_case

if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    for integ in (1, 0):
        for rough in (False, True):
            t0 = time.time()
            e_hip, e_o32, maps = grad_case(B, integ, rough, H=H)
            q = lambda t: ['%.1e' % v for v in np.percentile(t.numpy(), [5, 25, 50, 75, 95])]      # noqa: E731
            bar = torch.clamp(3.0 * e_o32, min=2e-4)
            print(dict(integ=integ, rough=rough, B=B, H=H, e_hip=q(e_hip), e_o32=q(e_o32), over_bar=float((e_hip > bar).float().mean()),
                       ratio_pct=q(e_hip / e_o32.clamp_min(1e-12)), maps={k: ('%.1e' % a, '%.1e' % b) for k, (a, b) in maps.items()}, seconds=round(time.time() - t0, 1)), flush=True)
