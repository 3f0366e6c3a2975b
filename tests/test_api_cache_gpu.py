"""The drop-in route at the fused step's speed (monoforce_amd/api_cache.py): `DPhysics.forward` + `monoforce.losses.physics_loss` + `loss.backward()`
as the reference's scripts write them (scripts/fit_terrain.py:53-62), replayed as ONE hipGraph after a few identical cycles.  Every test holds
the cached route to the SAME calls with the cache switched off (launch by launch), which the other test files hold to the oracle."""
import pytest
import torch

from tests.test_rollout_gpu import make_dphysics

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _problem(B=256, T=100, every=10, integ=1, seed=0):
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    dp = make_dphysics(pts, masks, integ, 0.05, 6.4)
    dp.dphys_cfg.traj_sim_time = 5.0
    z, mu = syn.bump_terrain(syn.bump_params(5 + seed), 6.4, 0.05).to(DEV), syn.wave_friction(6.4, 0.05).to(DEV)
    ctrl = syn.const_controls(B, T, seed=2 + seed).to(DEV)
    with torch.no_grad():
        (Xg, Xdg, Rg, Og), _ = dp(syn.bump_terrain(syn.bump_params(100), 6.4, 0.05).to(DEV).unsqueeze(0), ctrl, friction=mu.unsqueeze(0))
    full_ts = torch.linspace(0, 5.0, 500, device=DEV)[:T]
    sel = torch.arange(every - 1, T, every, device=DEV)
    states_gt = [t[:, sel].contiguous() for t in (Xg, Xdg, Rg, Og)]
    pred_ts, gt_ts = full_ts.unsqueeze(0).expand(B, -1), full_ts[sel].unsqueeze(0).expand(B, -1).contiguous()
    return dp, z, mu, ctrl, states_gt, pred_ts, gt_ts


def _fit(dp, z0, mu0, ctrl, states_gt, pred_ts, gt_ts, iters, enabled, extra=None, lr=0.05, gt_switch=None):
    """`iters` iterations of the reference's fit loop (a seeded in-place parameter update in place of Adam); returns per-iteration (loss, gz, gmu, Xs, Fs)."""
    from monoforce_amd import api_cache
    from monoforce.losses import physics_loss          # the reference's import path
    keep, api_cache.ENABLED = api_cache.ENABLED, enabled
    try:
        z, mu = z0.clone().unsqueeze(0).requires_grad_(True), mu0.clone().unsqueeze(0).requires_grad_(True)
        hist = []
        for i in range(iters):
            z.grad = mu.grad = None
            states, forces = dp(z_grid=z, controls=ctrl, friction=mu)
            gt = states_gt if gt_switch is None or i < gt_switch[0] else gt_switch[1]
            loss = physics_loss(states_pred=states, states_gt=gt, pred_ts=pred_ts, gt_ts=gt_ts, gamma=0.9)
            if extra is not None:
                loss = loss + extra(states, forces, i)
            loss.backward()
            hist.append((float(loss.detach()), z.grad.clone(), mu.grad.clone(), states[0].detach().clone(), forces[0].detach().clone()))
            with torch.no_grad():      # an in-place parameter update, as an optimizer makes it -- a seeded one, so that both loops see the same
                g = torch.Generator(device=DEV).manual_seed(100 + i)      # inputs in every iteration (a gradient step would feed the float atomics'
                z += lr * 0.02 * (torch.rand(z.shape, generator=g, device=DEV) - 0.5)      # rounding order back into the trajectory)
                mu += lr * 0.02 * (torch.rand(mu.shape, generator=g, device=DEV) - 0.5)
        cache = dp.__dict__.get('_api_step_cache')
        return hist, (cache.replays if cache is not None else 0)
    finally:
        api_cache.ENABLED = keep


def _same(a, b, tol=2e-5):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert abs(x[0] - y[0]) <= tol * abs(y[0]), (i, x[0], y[0])
        for k in (1, 2, 3, 4):
            scale = float(y[k].abs().max())
            assert float((x[k] - y[k]).abs().max()) <= tol * max(scale, 1e-30), (i, k, float((x[k] - y[k]).abs().max()), scale)


@pytest.mark.parametrize('integ', [1, 0])
@pytest.mark.parametrize('B', [256, 3072])
def test_cached_step_equals_launch_by_launch(integ, B):
    """Ten iterations of the fit loop: the first three run launch by launch, the capture happens in the fourth call, replays from there on --
    in-place parameter updates included (the graph reads the maps where they lie).  Loss, gradients, states and forces of EVERY iteration
    equal the launch-by-launch loop's (B = 3072: beyond the streaming backward, where the step's loss is not fused -- the cached step then
    replays the unfused sequence)."""
    args = _problem(B=B, integ=integ)
    ref, n0 = _fit(*args, iters=10, enabled=False)
    got, n1 = _fit(*_problem(B=B, integ=integ), iters=10, enabled=True)
    assert n0 == 0 and n1 >= 6, (n0, n1)
    _same(got, ref)


def test_previous_states_stay_intact_and_held_outputs_are_never_overwritten():
    from monoforce_amd import api_cache
    from monoforce.losses import physics_loss
    dp, z0, mu0, ctrl, states_gt, pred_ts, gt_ts = _problem()
    z, mu = z0.clone().unsqueeze(0).requires_grad_(True), mu0.clone().unsqueeze(0).requires_grad_(True)
    held, copies = [], []
    for i in range(9):
        z.grad = mu.grad = None
        states, forces = dp(z_grid=z, controls=ctrl, friction=mu)
        physics_loss(states_pred=states, states_gt=states_gt, pred_ts=pred_ts, gt_ts=gt_ts, gamma=0.9).backward()
        held.append(states[0]); copies.append(states[0].detach().clone())      # the user keeps EVERY iteration's positions
        with torch.no_grad():
            z -= 0.05 * z.grad
    cache = dp._api_step_cache
    assert 1 <= cache.replays <= 2, cache.replays          # two buffer sets, both still referenced: every later call ran launch by launch
    for h, c in zip(held, copies):
        assert torch.equal(h.detach(), c)
    assert float((copies[-1] - copies[0]).abs().max()) > 0  # (the terrain did change in between)
    del held, states, forces
    n = cache.replays
    for i in range(3):
        z.grad = mu.grad = None
        states, forces = dp(z_grid=z, controls=ctrl, friction=mu)
        physics_loss(states_pred=states, states_gt=states_gt, pred_ts=pred_ts, gt_ts=gt_ts, gamma=0.9).backward()
    assert cache.replays > n                                # released: the cached step runs again


def test_another_loss_on_the_cached_states_and_a_scaled_loss():
    """A second differentiable consumer of the handed-out states (here: of the velocities and the spring forces) sends a gradient into the
    cached step's own backward, which re-runs the rollout launch by launch; `physics_loss` itself still comes from the graph, scaled by 3."""
    def extra(states, forces, i):
        return 1e-3 * (states[1] ** 2).mean() + 1e-9 * (forces[0] ** 2).mean()
    from monoforce_amd import api_cache
    args = _problem(seed=1)
    ref, _ = _fit(*args, iters=7, enabled=False, extra=lambda s, f, i: 2.0 * 0 + extra(s, f, i))
    got, n = _fit(*_problem(seed=1), iters=7, enabled=True, extra=lambda s, f, i: 2.0 * 0 + extra(s, f, i))
    assert n >= 3
    _same(got, ref)


def test_another_ground_truth_leaves_the_cache_and_stays_correct():
    dp, z0, mu0, ctrl, states_gt, pred_ts, gt_ts = _problem(seed=2)
    other = [t + 0.05 for t in states_gt[:1]] + states_gt[1:]
    ref, _ = _fit(dp, z0, mu0, ctrl, states_gt, pred_ts, gt_ts, iters=12, enabled=False, gt_switch=(6, other))
    dp2, *rest = _problem(seed=2)
    got, n = _fit(dp2, *rest, iters=12, enabled=True, gt_switch=(6, other))
    assert n >= 4                                            # cached with the first ground truth, again (after three cycles) with the second
    _same(got, ref)


def test_inference_and_other_callers_are_untouched():
    dp, z0, mu0, ctrl, states_gt, pred_ts, gt_ts = _problem()
    with torch.no_grad():
        for _ in range(6):
            dp(z_grid=z0.unsqueeze(0), controls=ctrl, friction=mu0.unsqueeze(0))
    assert dp._api_step_cache.replays == 0 and dp._api_step_cache.entry is None


def test_every_row_stamped_like_fit_terrain_and_non_contiguous_inputs_stay_launch_by_launch():
    """`scripts/fit_terrain.py:53-62` passes `gt_ts = pred_ts` (every output row carries a stamp): cached, equal to launch by launch.  Controls
    that are a strided view of a larger tensor would have to be compacted into a copy first -- the graph would read that copy for ever: such
    calls never arm the cache."""
    args = _problem(B=64, T=60, every=1, seed=3)
    ref, _ = _fit(*args, iters=8, enabled=False)
    got, n = _fit(*_problem(B=64, T=60, every=1, seed=3), iters=8, enabled=True)
    assert n >= 4
    _same(got, ref)
    dp, z0, mu0, ctrl, states_gt, pred_ts, gt_ts = _problem(B=64, T=60, seed=4)
    wide = torch.zeros(64, 60, 4, device=DEV)
    wide[..., :2] = ctrl
    strided = wide[..., :2]                                   # same values, rows 16 bytes apart
    assert not strided.is_contiguous()
    ref, _ = _fit(dp, z0, mu0, ctrl, states_gt, pred_ts, gt_ts, iters=7, enabled=False)
    dp2, *_ = _problem(B=64, T=60, seed=4)
    got, n = _fit(dp2, z0, mu0, strided, states_gt, pred_ts, gt_ts, iters=7, enabled=True)
    assert n == 0
    _same(got, ref)


def test_capture_holds_the_garbage_collector_off():
    """monoforce_amd/capture.py: a hipGraph lying in a reference cycle (a discarded cache entry) is collected BEFORE a new capture begins and the
    cyclic collector stays off until it has ended -- torch 2.10 no longer collects in `torch.cuda.graph.__enter__`, and a collection inside a
    capture that destroys a graph aborts the process (`~CUDAGraph`: "operation not permitted when stream is capturing";
    tools/debug_gc_capture.py reproduces it).  `pytest tests -m gpu -q` died of exactly that in the re-capture of
    test_another_ground_truth_leaves_the_cache_and_stays_correct, by the collector's timing."""
    import gc
    import weakref
    from monoforce_amd.capture import capture
    x = torch.zeros(256, device=DEV)
    s = torch.cuda.Stream()

    class Holder:
        pass
    h = Holder(); h.me = h
    h.g = torch.cuda.CUDAGraph()
    with capture(h.g, stream=s, capture_error_mode='thread_local'):
        h.y = x * 2
    wr = weakref.ref(h)
    del h
    assert wr() is not None and gc.isenabled()
    g = torch.cuda.CUDAGraph()
    with capture(g, stream=s, capture_error_mode='thread_local'):
        assert wr() is None and not gc.isenabled()
        z = x + 1
    assert gc.isenabled()
    g.replay()
    torch.cuda.synchronize()
    assert float(z.min()) == 1.0
    # the collector's state is put back as it was found, also when the capture body raises
    gc.disable()
    try:
        with pytest.raises(ZeroDivisionError):
            with capture(torch.cuda.CUDAGraph(), stream=s, capture_error_mode='thread_local'):
                1 / 0
        assert not gc.isenabled()
    finally:
        gc.enable()
