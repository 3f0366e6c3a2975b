"""How far does the float32 fast-math rollout FOLLOW the float64 oracle, against how far the oracle's own float32 run does?  ("what's
weak": the float32 bars hold on the calm prefix; beyond it the system is chaotic and only boundedness is asserted.)  At the BASELINE batch:
1024 rollouts x 500 steps x 4 points on the headline's terrain (and a rougher one), both integrators -- per rollout the HORIZON = the first
step at which the positions leave the float64 oracle's by more than `tol` of the rollout's largest coordinate:

    horizon_hip[b]   HIP float32 fast math            vs   oracle float64
    horizon_o32[b]   oracle float32 (torch CPU ops)   vs   oracle float64

If the HIP arithmetic were worse than "another float32 evaluation order", its horizons would be systematically shorter.
    python tools/horizon_stats.py [B]          (test infrastructure: imports oracle/)"""
import os, sys
os.environ.setdefault('OMP_NUM_THREADS', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(min(int(os.environ['OMP_NUM_THREADS']), torch.get_num_threads()))
from tests.horizon_cases import case

if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # contact points (4: the BASELINE body; 175: the size of the reference's tradr body)
    for integ in (1, 0):
        for rough in (False, True):
            for what in ('Xs', 'Fs'):
                for tol in (1e-4, 1e-3):
                    print(case(B, integ, rough, tol, what, N), flush=True)
