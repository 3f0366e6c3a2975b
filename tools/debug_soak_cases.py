import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import soak_cases as sc, helpers as hp
kind, seed = sys.argv[1], int(sys.argv[2])
c = sc.build(kind, seed)
print({k: getattr(c, k) for k in ('B', 'T', 'H', 'res', 'N', 'integ', 'shared', 'all_outputs')}, getattr(c, 'where', None))
g = sc.run_hip(c); r64 = sc.run_oracle(c, torch.float64); r32 = sc.run_oracle(c, torch.float32)
print(g['kernel'])
for k in ('gz', 'gmu', 'gc'):
    if r64[k] is not None: print(k, 'hip %.3e oracle32 %.3e' % (hp.rel_err(g[k], r64[k]), hp.rel_err(r32[k], r64[k])))
dX = (g['Xs'].double() - r64['Xs']).abs().flatten(1).amax(1); dX32 = (r32['Xs'].double() - r64['Xs']).abs().flatten(1).amax(1)
sc_gc = float(r64['gc'].abs().max())
for b in range(c.sel.numel()):
    line = 'rollout %2d dXs hip %.2e oracle32 %.2e  gc err %.2e (o32 %.2e)' % (b, float(dX[b]), float(dX32[b]), float((g['gc'][b].double() - r64['gc'][b]).abs().max()) / sc_gc, float((r32['gc'][b].double() - r64['gc'][b]).abs().max()) / sc_gc)
    if r64['gz'].shape[0] == c.sel.numel() and c.sel.numel() > 1:
        s = float(r64['gz'].abs().max())
        line += '  gz[b] err %.2e (o32 %.2e)' % (float((g['gz'][b].double() - r64['gz'][b]).abs().max()) / s, float((r32['gz'][b].double() - r64['gz'][b]).abs().max()) / s)
    print(line)
