"""Every soak problem that landed outside the oracle-derived bar ("explained in text files, not pinned by tests"): round 5's three
-- LDS-window problem 53 (`tools/soak_win.py`, 357 x the bar), multi-wave problems 118 and 146 (`tools/soak_r5.py`, 1.4 x and 10.8 x) -- and the
sixteen of round 6's three further ranges of 928 problems each (SOAK_SEED0=2000 / 3000 / 4000, profiles/r6_soak_third_range.txt; 19 of 3712
problems in all, among them ONE of the component-parallel kernels).  Each is held to exactly its explanation, so
that a kernel that breaks one of them for a REAL reason is not waved through as "the known one":

  (a) the error sits in at most two rollouts -- found through the control gradients and per-rollout map gradients or, on ONE shared map pair
      whose control gradients do not show it, by attributing the map gradients rollout by rollout (the SAME launch with the loss restricted to
      one rollout at a time).  With those rollouts switched out of the loss, every gradient of the problem is within the ordinary bar
      max(2e-4, 3 x the oracle's own float32-vs-float64 distance);
  (b) the same kernels' control flow and indexing are right on those very rollouts: the float64 HIP build agrees with the float64 oracle;
  (c) it is not the kernel family: the other float32 route (register accumulators + atomics for the LDS window: MF_BWD_WIN=0; the general
      backward for the record-reading multi-wave kernels: MF_MW_BWD=0; one point per lane for the component-parallel kernels) gives the same
      float32 gradients -- or, where the event is narrower than the difference between two kernels' roundings, lands equally far from the
      oracle; where it lands at the ORACLE'S side (what a bug of this route would look like), (d) must find a kink of kind 3 and show it causal;
  (d) it IS a float32 event, one of five kinds:
      1. the float32 positions follow the float64 oracle's up to a step, then PART -- and around that step a contact point of the oracle's
         trajectory lies on a cell edge or outside the map (clamped indices);
      2. float32 cannot referee the rollout at all: the oracle's own float32 gradient is > 1 % off on it;
      3. the positions never part and the oracle's float32 gradient is right: a KINK within float32 resolution -- a contact point within a few
         float32 ulps of a cell edge (the interpolant is continuous there, its slopes and the cells the gradient lands in are not), or an
         unclamped force / angular acceleration within 1e-4 of its clamp (value continuous, derivative 1 or 0) -- causal: the same problem
         stopped in front of that row has an ordinary gradient;
      4. none of these: a long cancellation in the clamp cell (off-map contact points all deposit into the last cell), where every float32
         evaluation order is 1e-3 .. 1e-2 off -- asserted as measured: confined to that cell, the oracle's own float32 >= 1e-3 off there, the
         kernels within 5 x (precise=True) / 20 x (fast math) of it on the pinned sample;
      5. (cp 5788 alone) a kink the list does not name: localized to ONE row by bisection on the horizon, gone when the controls are scaled
         by 1 + 1e-5, crossed by this route's rounding only, the float64 build of the same kernels exact."""
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


@pytest.mark.parametrize('kind,seed,route_env', [('win', 53, 'MF_BWD_WIN'), ('mw', 118, 'MF_MW_BWD'), ('mw', 146, 'MF_MW_BWD'),
                                                 # a third range of problems (SOAK_SEED0=2000: 128 + 400 + 400, profiles/r6_soak*_seeds2000.txt)
                                                 ('win', 2054, 'MF_BWD_WIN'), ('mw', 2006, 'MF_MW_BWD'), ('mw', 2035, 'MF_MW_BWD'), ('mw', 2149, 'MF_MW_BWD'),
                                                 ('mw', 2340, 'MF_MW_BWD'),
                                                 # ... and a fourth (SOAK_SEED0=3000, profiles/r6_soak*_seeds3000.txt)
                                                 ('win', 3060, 'MF_BWD_WIN'), ('mw', 3102, 'MF_MW_BWD'), ('mw', 3164, 'MF_MW_BWD'), ('mw', 3320, 'MF_MW_BWD'),
                                                 ('mw', 3369, 'MF_MW_BWD'),
                                                 # ... and a fifth (SOAK_SEED0=4000): the first outlier of the component-parallel kernels among 2000 problems
                                                 ('cp', 4288, 'MF_SOAK_CP_LANES'), ('mw', 4036, 'MF_MW_BWD'), ('mw', 4067, 'MF_MW_BWD'), ('mw', 4200, 'MF_MW_BWD'),
                                                 ('mw', 4275, 'MF_MW_BWD'), ('mw', 4382, 'MF_MW_BWD'),
                                                 # ... and 1000 more problems of the component-parallel kernels alone (SOAK_SEED0=5000, profiles/r6_soak_cp_seeds5000.txt)
                                                 ('cp', 5788, 'MF_SOAK_CP_LANES'), ('cp', 5910, 'MF_SOAK_CP_LANES')])
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
    single = {}
    if not bad and max(ratios.values()) > 1.0 and len(measures) == 1:
        # one shared map pair and the control gradients do not show it: attribute the map gradients rollout by rollout -- the SAME launch with the
        # loss restricted to one rollout at a time, against the oracle run on that rollout alone
        for k in range(n):
            single[k] = sc.single_rollout_errors(c, k)
            floor = 1e-4 if kind == 'win' else 1e-3      # (half the floor of the soak that found the problem: tools/soak_win.py 2e-4, soak_r5.py 2e-3)
            per[k] = max(eh / max(3.0 * eo, floor) for eh, eo in single[k].values())
            unrefereed[k] = any(eo > 1e-2 for _, eo in single[k].values())
        # (a rollout float32 cannot referee counts too: its own float32 error is what lifts the problem's gradient off the float64 one, whichever
        #  float32 evaluation order runs)
        bad = [k for k in range(n) if per[k] > 1.0 or bool(unrefereed[k])]
        print('attributed through the map gradients:', {k: round(float(per[k]), 2) for k in bad})
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
    g64 = sc.run_hip(c, torch.float64, rows=rows, points_per_lane=16 if kind in ('mw', 'cp') else 0)
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
    needs_kink = False
    for k in NAMES:
        if g[k] is None:
            continue
        e_here, e_there = hp.rel_err(g[k], r64[k]), hp.rel_err(other[k], r64[k])
        d = hp.rel_err(other[k], g[k])
        if bool(unrefereed[bad].any()) or d > max(2e-3 * e_here, 5e-5):
            # a rollout whose gradient float32 cannot referee (the oracle's own float32 run is > 1 % off on it), or an event narrower than the
            # difference between two kernels' roundings: every float32 evaluation order lands somewhere else -- the other route is as far from
            # the oracle as this one, not closer (a bug of THIS route would leave the other one at the oracle's side)
            # ... unless the event is a kink within float32 resolution that only ONE route's rounding crosses (cp 4288: |wd| within 2.4e-7 of
            # omega_max -- the component-parallel kernels cross it, the one-point-per-lane kernels and the oracle's float32 do not): then (d) must
            # find that kink and show it causal, whatever the positions do
            if e_there < 0.1 * e_here:
                needs_kink = True
            else:
                assert e_there <= 10 * max(e_here, 1e-6), (k, e_here, e_there, d)
        # (else: same arithmetic per contribution, another order of the float atomics -- the two routes agree far below their distance to the oracle)
    # (d) the event: float32 follows float64 up to a step, and a contact point of the float64 trajectory is then on an edge of its cell,
    # off the map, or at a switch
    P = torch.as_tensor(c.pts, dtype=torch.float64)
    for k in bad:
        dX = (g['Xs'][k].double() - r64['Xs'][k]).abs().amax(-1)                   # [T]
        parted = torch.nonzero(dX > max(5 * float(dX[:max(c.T // 8, 2)].max()), 2e-6)).flatten()
        if parted.numel() and not needs_kink:   # the float32 trajectory FOLLOWED the oracle's, then parted at t_star
            t_star = int(parted[0])
            assert float(dX[:max(t_star - 1, 1)].max()) <= 2e-5, (k, t_star, float(dX[:max(t_star - 1, 1)].max()))
            lo, hi = max(t_star - 6, 0), min(t_star + 2, c.T)
        elif bool(unrefereed[k]) and not needs_kink:      # no visible parting, and float32 cannot referee the rollout's gradient at all (checked above)
            lo, hi = 0, c.T
        else:
            # no visible parting and the oracle's float32 gradient of THIS input is fine: a discontinuity of the GRADIENT alone.  The interpolant is
            # continuous across a cell edge, its slopes (the normal, the cells the gradient is deposited in) are not: a contact point of the
            # float64 trajectory within a few float32 ulps of an edge is IN one cell or the other by the rounding of one addition -- the
            # oracle's float32 run lands on the float64 side, the kernels on the other (mw 2006: 2.5e-7 cells from an edge at step 56, where a
            # float32 ulp of the cell coordinate is 3.8e-6; of three float32 builds of the kernels -- fast math, exact division / square root,
            # exact reciprocal norms -- one lands there).  No displacement or reordering of the oracle reaches a window that narrow; what can
            # be checked is causality: such a point exists, and the SAME problem stopped in front of it has an ordinary gradient.  The clamps on
            # the forces and on the angular acceleration (dphysics.py:233,250-251,257) are kinks of the same nature -- value continuous,
            # derivative 1 or 0 (win 3060: |wd| within 1.8e-6 of omega_max at step 31) -- and are looked for alongside (sc.kink_rows)
            kinks = sc.kink_rows(c, k)
            R, X = r64['Rs'][k], r64['Xs'][k]
            u = (((X - R[:, :, 2] * sc.SINK).unsqueeze(1) + torch.einsum('tij,nj->tni', R, P))[..., :2] + c.d_max) / c.res          # [T, N, 2]
            rows_hit = torch.tensor([t for t, _, _ in kinks])
            if not rows_hit.numel():
                # no such point either.  What is left (mw 2035, a robot that starts OFF the map): every off-map contact point deposits its height
                # gradient in the LAST cell (the reference clamps the flat index, dphysics.py:427-430), step after step with alternating sign --
                # the cell's value is what is left of a long cancellation, and EVERY float32 evaluation is 1e-3 .. 1e-2 off there: over 24
                # control sequences of this rollout (tools/clamp_cell_noise.py, profiles/r6_clamp_cell_noise.txt) the medians are 2.0e-3 for
                # the oracle's float32, 3.6e-3 for the IEEE float32 kernels (precise=True), 6.0e-3 for the fast-math kernels, sample by sample
                # anywhere between 0.1 x and 30 x each other -- and a build of the fast kernels with every approximation replaced by its IEEE
                # operation gives 2.36e-2 where they give 2.33e-2: it is the evaluation order, not the approximations.  Held to that: the
                # difference sits in the clamp cell, the oracle's own float32 is >= 1e-3 off there, the kernels within 5 x (IEEE) / 20 x
                # (fast) of it on this sample (measured: 1.8 x / 10 x), the other gradients within the ordinary bar
                e_fast, dmap = sc.single_rollout_errors(c, k, with_diff=True)
                dmap = dmap.abs()[0]
                if not (bool((u > c.H - 1).any() | (u < 0).any()) and float(dmap[-1, -1]) >= 0.5 * float(dmap.max())):
                    # not that either (cp 5788, the one such case among 3000 problems of the component-parallel kernels): a kink this file's
                    # list of kinks does not know.  What can be held without naming it: it is ONE row's event (the problem stopped in front of
                    # that row -- found by bisection on the horizon -- has an ordinary gradient), it is narrower than 1e-5 relative in the
                    # controls (scaled by 1 + 1e-5 the full problem has an ordinary gradient), only this route's rounding crosses it (the
                    # other float32 route was at the oracle's side: `needs_kink`), and the float64 build of the same kernels agrees with the
                    # float64 oracle ((b) above)
                    assert needs_kink, (k, 'unexplained', e_fast)
                    bar = lambda e: all(v[0] <= max(3.0 * v[1], 1e-4) for v in e.values())      # noqa: E731
                    lo_t, hi_t = 4, c.T
                    with sc.truncated(c, lo_t) as ct:
                        assert bar(sc.single_rollout_errors(ct, k)), (k, 'off from the first rows')
                    while hi_t - lo_t > 1:
                        mid = (lo_t + hi_t) // 2
                        with sc.truncated(c, mid) as ct:
                            ok_mid = bar(sc.single_rollout_errors(ct, k))
                        lo_t, hi_t = (mid, hi_t) if ok_mid else (lo_t, mid)
                    keep_ctrl = c.ctrl
                    try:
                        c.ctrl = (keep_ctrl.double() * (1.0 + 1e-5)).float()
                        e_moved = sc.single_rollout_errors(c, k)
                    finally:
                        c.ctrl = keep_ctrl
                    print('rollout', k, 'an event of ONE row the list of kinks does not name: ordinary gradient up to', lo_t, 'rows, off from', hi_t,
                          '| full horizon', {a: '%.1e' % v[0] for a, v in e_fast.items()}, '| controls x (1 + 1e-5)', {a: '%.1e' % v[0] for a, v in e_moved.items()})
                    assert bar(e_moved), (k, e_moved)
                    assert max(v[0] for v in e_fast.values()) <= 5e-2, (k, e_fast)
                    continue
                e_prec = sc.single_rollout_errors(c, k, precise=True)
                print('rollout', k, 'cancellation in the clamp cell | own gz error: fast %.1e, precise %.1e, oracle float32 %.1e' % (e_fast['gz'][0], e_prec['gz'][0], e_fast['gz'][1]))
                assert e_fast['gz'][1] >= 1e-3, (k, e_fast['gz'])                       # ... which the oracle's own float32 run shows as well
                assert e_prec['gz'][0] <= max(5.0 * e_prec['gz'][1], 1e-3), (k, e_prec['gz'])
                assert e_fast['gz'][0] <= 20.0 * e_fast['gz'][1], (k, e_fast['gz'])
                for key in ('gmu', 'gc'):
                    if key in e_fast:
                        assert e_fast[key][0] <= max(3.0 * e_fast[key][1], 1e-4), (k, key, e_fast[key])
                continue
            e_full = single.get(k) or sc.single_rollout_errors(c, k)
            # the kinks nearest to float32 resolution first (edges: ulps / 4, clamps: relative margin / 2e-5); the one that is causal: the problem
            # stopped in front of it has an ordinary gradient (and is long enough to be a problem at all: >= 4 rows)
            ranked = sorted(kinks, key=lambda q: q[2] / (4.0 if 'edge' in q[1] else 1e-4))
            found = None
            for t_kink, what, margin in [q for q in ranked if q[0] >= 4][:4]:
                with sc.truncated(c, t_kink) as ct:
                    e_cut = sc.single_rollout_errors(ct, k)
                if all(v[0] <= max(3.0 * v[1], 1e-4) for v in e_cut.values()):
                    found = (t_kink, what, margin, e_cut)
                    break
            if found is None:
                # every candidate sits in the first rows: nothing to stop in front of.  Then the kink must be sharper than one ulp / 5e-6
                early = [q for q in ranked if q[0] < 4 and q[2] <= (1.0 if 'edge' in q[1] else 5e-6)]
                assert early, (k, 'no kink whose removal cures the gradient', ranked[:4])
                print('rollout', k, 'on a kink within float32 resolution in its first rows:', early[0], '| own gradient errors (HIP, oracle float32)',
                      {a: ('%.1e' % v[0], '%.1e' % v[1]) for a, v in e_full.items()})
                continue
            print('rollout', k, 'on a kink within float32 resolution:', found[:3], '| its own gradient errors (HIP, oracle float32): full horizon',
                  {a: ('%.1e' % v[0], '%.1e' % v[1]) for a, v in e_full.items()}, 'stopped in front of it', {a: ('%.1e' % v[0], '%.1e' % v[1]) for a, v in found[3].items()})
            continue
        R, X = r64['Rs'][k], r64['Xs'][k]                                          # [T,3,3], [T,3]
        x = X - R[:, :, 2] * sc.SINK
        p = x.unsqueeze(1) + torch.einsum('tij,nj->tni', R, P)                      # [T,N,3] contact points of the float64 trajectory
        u = (p[..., :2] + c.d_max) / c.res                                          # cell coordinates (dphysics.py:419-420)
        frac = u[lo:hi] - torch.floor(u[lo:hi])
        edge = torch.minimum(frac, 1 - frac).min()
        off_map = bool(((u[lo:hi] < 0) | (u[lo:hi] > c.H - 1)).any())
        # (a float32 cell coordinate of magnitude ~H carries ~H * 6e-8 of rounding; the step moves a point by ~2e-3 cells at most)
        assert float(edge) <= 4e-3 or off_map, (k, lo, hi, float(edge), off_map)
