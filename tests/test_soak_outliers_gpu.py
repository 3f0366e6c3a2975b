"""The three soak problems outside the oracle-derived bar ("explained in text files, not pinned by tests"): LDS-window problem 53
(`tools/soak_win.py`, 357 x the bar) and multi-wave problems 118 and 146 (`tools/soak_r5.py`, 1.4 x and 10.8 x).  Each is held to exactly the
explanation the profiles give -- so that a kernel that breaks them for a REAL reason is not waved through as "the known one":

  (a) the error sits in at most two rollouts: with those rollouts switched out of the loss, every gradient of the problem is within the
      ordinary bar max(2e-4, 3 x the oracle's own float32-vs-float64 distance);
  (b) the same kernels' control flow and indexing are right on those very rollouts: the float64 HIP build agrees with the float64 oracle;
  (c) it is not the kernel family: the other float32 route (register accumulators + atomics for the LDS window: MF_BWD_WIN=0; the general
      backward for the record-reading multi-wave kernels: MF_MW_BWD=0) gives the same float32 gradients;
  (d) it IS a float32 event of the trajectory: up to some step the float32 HIP positions of the rollout follow the float64 oracle's, then
      they part -- and around that step a contact point of the oracle's trajectory lies on a cell edge (within a few float32 ulps of the cell
      coordinate), outside the map (clamped indices), or at the soft contact switch / a clamp where one ulp decides the branch."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

from tests import helpers as hp
from tests import soak_cases as sc

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ('gz', 'gmu', 'gc')


def _bars(r32, r64):
    return {k: max(2e-4, 3.0 * hp.rel_err(r32[k], r64[k])) for k in NAMES if r64[k] is not None}


def _per_rollout_error(g, r64):
    scale = float(r64['gc'].abs().max())
    return (g['gc'].double() - r64['gc']).abs().flatten(1).amax(1) / scale


@pytest.mark.parametrize('kind,seed,route_env', [('win', 53, 'MF_BWD_WIN'), ('mw', 118, 'MF_MW_BWD'), ('mw', 146, 'MF_MW_BWD')])
def test_soak_outlier_is_the_float32_event_the_profiles_describe(kind, seed, route_env):
    c = sc.build(kind, seed)
    g = sc.run_hip(c)
    r64, r32 = sc.run_oracle(c, torch.float64), sc.run_oracle(c, torch.float32)
    bars = _bars(r32, r64)
    ratios = {k: hp.rel_err(g[k], r64[k]) / bars[k] for k in bars}
    # the problem as the soak scores it (kept as a record of what is being explained; no assertion on `> 1`: a future kernel may well land
    # on the oracle's side of the event)
    print(kind, seed, 'error / bar', {k: round(v, 2) for k, v in ratios.items()}, g['kernel'][-70:])
    # per-rollout errors: through the control gradient of each selected rollout and, where every rollout has its own maps, through its own map
    # gradients.  A rollout is an outlier when the HIP error exceeds three times the oracle's own float32 error on it (floor 1e-4) -- or when
    # the oracle's float32 run is ITSELF more than 1 % off on it: float32 cannot referee that rollout at all
    n = c.sel.numel()
    measures = [('gc', float(r64['gc'].abs().max()))]
    if r64['gz'].shape[0] == n and n > 1:
        measures += [(k, float(r64[k].abs().max())) for k in ('gz', 'gmu') if r64[k] is not None]
    score, unrefereed = torch.zeros(n), torch.zeros(n, dtype=torch.bool)
    for k, scale in measures:
        e_hip = (g[k].double() - r64[k]).abs().flatten(1).amax(1) / scale
        e_o32 = (r32[k].double() - r64[k]).abs().flatten(1).amax(1) / scale
        score = torch.maximum(score, e_hip / torch.clamp(3.0 * e_o32, min=1e-4))
        unrefereed |= e_o32 > 1e-2
    bad = [int(k) for k in torch.nonzero((score > 1.0) | unrefereed).flatten()]
    per = score
    assert len(bad) <= 2, (bad, per[bad])
    if not bad:
        assert max(ratios.values()) <= 1.0, ratios       # nothing to explain: then the problem must simply pass
        return
    # (a) without those rollouts the problem passes at the ordinary bar
    mask = torch.ones(n); mask[bad] = 0.0
    g_m, r64_m, r32_m = sc.run_hip(c, rows_mask=mask), sc.run_oracle(c, torch.float64, rows_mask=mask), sc.run_oracle(c, torch.float32, rows_mask=mask)
    for k, bar in _bars(r32_m, r64_m).items():
        assert hp.rel_err(g_m[k], r64_m[k]) <= bar, (k, hp.rel_err(g_m[k], r64_m[k]), bar)
    # (b) the float64 build on those rollouts (and two healthy neighbours) against the float64 oracle
    rows = torch.tensor(sorted(set(bad + [0, n - 1])))      # positions in the selection
    # (win: the exact float64 instantiation of the same `rollout_bwd_body`; mw: the float64 validation build of the record-reading kernels)
    g64 = sc.run_hip(c, torch.float64, rows=rows, points_per_lane=16 if kind == 'mw' else 0)
    o64 = sc.run_oracle(c, torch.float64, rows=rows)
    for k in NAMES:
        if o64[k] is not None:
            assert hp.rel_err(g64[k], o64[k]) <= 1e-7, (k, hp.rel_err(g64[k], o64[k]))
    # (c) the other float32 route gives the same gradients (its environment switch is read once per process: a child)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'g.pt')
        r = subprocess.run([sys.executable, '-m', 'tests.soak_cases', kind, str(seed), out], env=dict(os.environ, **{route_env: '0'}), cwd=REPO,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        other = torch.load(out)
    assert other['kernel'] != g['kernel'], (other['kernel'], g['kernel'])
    for k in NAMES:
        if g[k] is None:
            continue
        e_here, e_there = hp.rel_err(g[k], r64[k]), hp.rel_err(other[k], r64[k])
        if bool(unrefereed[bad].any()):
            # a rollout whose gradient float32 cannot referee (the oracle's own float32 run is > 1 % off on it): every float32 evaluation order
            # lands somewhere else -- the other route is as far from the oracle as this one, not closer
            assert 0.1 * e_here <= e_there <= 10 * max(e_here, 1e-6), (k, e_here, e_there)
        else:
            # same arithmetic per contribution, another order of the float atomics: the two routes agree far below their distance to the oracle
            d = hp.rel_err(other[k], g[k])
            assert d <= max(2e-3 * e_here, 5e-5), (k, d, e_here)
    # (d) the event: float32 follows float64 up to a step, and a contact point of the float64 trajectory is then on an edge of its cell,
    # off the map, or at a switch
    P = torch.as_tensor(c.pts, dtype=torch.float64)
    for k in bad:
        dX = (g['Xs'][k].double() - r64['Xs'][k]).abs().amax(-1)                   # [T]
        parted = torch.nonzero(dX > max(5 * float(dX[:max(c.T // 8, 2)].max()), 2e-6)).flatten()
        if parted.numel():                      # the float32 trajectory FOLLOWED the oracle's, then parted at t_star
            t_star = int(parted[0])
            assert float(dX[:max(t_star - 1, 1)].max()) <= 2e-5, (k, t_star, float(dX[:max(t_star - 1, 1)].max()))
            lo, hi = max(t_star - 6, 0), min(t_star + 2, c.T)
        else:                                   # no visible parting: then float32 must be unable to referee the rollout's gradient (checked above)
            assert bool(unrefereed[k]), (k, float(dX.max()))
            lo, hi = 0, c.T
        R, X = r64['Rs'][k], r64['Xs'][k]                                          # [T,3,3], [T,3]
        x = X - R[:, :, 2] * sc.SINK
        p = x.unsqueeze(1) + torch.einsum('tij,nj->tni', R, P)                      # [T,N,3] contact points of the float64 trajectory
        u = (p[..., :2] + c.d_max) / c.res                                          # cell coordinates (dphysics.py:419-420)
        frac = u[lo:hi] - torch.floor(u[lo:hi])
        edge = torch.minimum(frac, 1 - frac).min()
        off_map = bool(((u[lo:hi] < 0) | (u[lo:hi] > c.H - 1)).any())
        # (a float32 cell coordinate of magnitude ~H carries ~H * 6e-8 of rounding; the step moves a point by ~2e-3 cells at most)
        assert float(edge) <= 4e-3 or off_map, (k, lo, hi, float(edge), off_map)
