"""Per-iteration differences of the cached drop-in step against launch by launch, with a second loss on the states (tests/test_api_cache_gpu.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_api_cache_gpu import _problem, _fit
which = sys.argv[1] if len(sys.argv) > 1 else 'xds'
extras = {'xds': lambda s, f, i: 1e-3 * (s[1] ** 2).mean(), 'fs': lambda s, f, i: 1e-9 * (f[0] ** 2).mean(), 'none': None,
          'xs': lambda s, f, i: 1e-3 * (s[0] ** 2).mean()}
ref, _ = _fit(*_problem(seed=1), iters=9, enabled=False, extra=extras[which])
got, n = _fit(*_problem(seed=1), iters=9, enabled=True, extra=extras[which])
print('replays', n)
for i, (a, b) in enumerate(zip(got, ref)):
    print(i, 'loss %.3e' % (abs(a[0] - b[0]) / abs(b[0])), ' '.join('%s %.2e' % (nm, float((a[k] - b[k]).abs().max()) / float(b[k].abs().max())) for k, nm in ((1, 'gz'), (2, 'gmu'), (3, 'Xs'), (4, 'Fs'))))
