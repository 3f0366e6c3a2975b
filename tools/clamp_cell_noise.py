"""mw 2035 (a robot that starts off the map: every off-map contact point deposits its height gradient in the LAST cell, a long cancellation): is
the fast-math kernels' error there a property of fast math, or one draw from the float32 noise every evaluation order has?  The same problem
with 24 other control sequences for rollout 9: its own map-gradient error -- fast-math kernels, IEEE float32 kernels (precise=True), the
oracle's float32 -- against the float64 oracle.    python tools/clamp_cell_noise.py [n]"""
import os, sys
os.environ.setdefault('OMP_NUM_THREADS', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(8)
from monoforce_amd import synthetic as syn
from tests import soak_cases as sc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
c = sc.build('mw', 2035)
k = 9
keep = c.ctrl.clone()
rows = []
for j in range(n):
    c.ctrl = keep.clone()
    if j:
        c.ctrl[c.sel[k]] = syn.varying_controls(1, c.T, seed=5000 + j, dtype=torch.float64).float()[0]
    ef = sc.single_rollout_errors(c, k)
    ep = sc.single_rollout_errors(c, k, precise=True)
    rows.append((ef['gz'][0], ep['gz'][0], ef['gz'][1]))
    print(j, 'fast %.2e  precise %.2e  oracle float32 %.2e' % rows[-1], flush=True)
a = np.array(rows)
print('medians: fast %.2e  precise %.2e  oracle float32 %.2e' % tuple(np.median(a, 0)))
print('90th percentiles: fast %.2e  precise %.2e  oracle float32 %.2e' % tuple(np.percentile(a, 90, 0)))
print('fast / oracle ratio: median %.2f, max %.2f;  precise / oracle: median %.2f, max %.2f' % (np.median(a[:, 0] / a[:, 2]), (a[:, 0] / a[:, 2]).max(), np.median(a[:, 1] / a[:, 2]), (a[:, 1] / a[:, 2]).max()))
