"""Follow-up of tools/ab_step_fwd.py: the forward of a saturated fit step takes 0.56 ms where the same kernel alone takes 0.37 (16 384 rollouts).
Which neighbour does it?  Sequences of the step's own launches with HIP events around every forward:
  A  fwd fwd fwd                      (alone)
  B  [fwd value bwd reduce] fwd fwd   (two more forwards right behind a whole step: is only the FIRST one slow?)
  C  step, host sleep 20 ms, fwd      (an idle gap: clocks / power state)
  D  step with the backward's rows COPIED first (the backward reads other buffers than the next forward writes)
AB_B=16384 python tools/ab_step_fwd2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd import _timing, synthetic as syn
from monoforce_amd.train import TerrainFitProblem
DEV = 'cuda'
B = int(os.environ.get('AB_B', '16384'))
cfg, dp, pts, masks, z, mu, ctrl = build_problem(B, 500, 4, DEV, 1)
cd = ctrl.to(DEV)
zl, ml = z.to(DEV).clone().requires_grad_(True), mu.to(DEV).clone().requires_grad_(True)
prob = TerrainFitProblem(dp, syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(DEV), mu.to(DEV), cd)


def fwd():
    dp.return_forces = False
    try:
        return dp(zl.unsqueeze(0), cd, friction=ml.unsqueeze(0))
    finally:
        dp.return_forces = True


def fwd_times(seq, n=4):
    for _ in range(2):
        seq()
    torch.cuda.synchronize()
    _timing.start()
    for _ in range(n):
        seq()
    k = _timing.stop()
    f = k['rollout_fwd_kernel']
    per = len(f) // n
    return [round(float(np.mean(f[i::per])), 4) for i in range(per)], {a: round(float(np.mean(v)), 4) for a, v in k.items() if a != 'rollout_fwd_kernel'}


def seq_a():
    fwd(); fwd(); fwd()


def seq_b():
    prob.step(zl, ml, eager=True); fwd(); fwd()


def seq_c():
    prob.step(zl, ml, eager=True); torch.cuda.synchronize(); time.sleep(0.02); fwd()


big = torch.empty(256 * 1024 * 1024 // 4, device=DEV)


def seq_e():      # a 256 MB fill between the step and the next forward (flushes L2 / the memory-side cache of the backward's lines)
    prob.step(zl, ml, eager=True); big.zero_(); fwd()


def seq_f():      # the controls (65 MB at 16 384 x 500) read once between the step and the forward: back in the memory-side cache?
    prob.step(zl, ml, eager=True); cd.sum(); fwd()


def seq_g():      # ... and the other way round: the fill evicts them in front of an otherwise 'alone' forward
    fwd(); big.zero_(); fwd(); cd.sum(); fwd()


for name, s in (('F step read-controls fwd', seq_f), ('G fwd fill fwd read-controls fwd', seq_g), ('A fwd fwd fwd', seq_a), ('B step fwd fwd', seq_b), ('C step sleep fwd', seq_c), ('E step fill256MB fwd', seq_e)):
    f, rest = fwd_times(s)
    print(f'B={B} {name:24s} forwards in order {f}  others {rest}', flush=True)
