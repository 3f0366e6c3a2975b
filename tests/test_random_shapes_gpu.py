"""Seeded random problem shapes: HIP (float64, through the C ABI) vs the CPU oracle, forward and gradients.

The fixed-shape tests pin the reference's numbers; this sweep varies what they hold constant -- batch size, horizon, number
of contact points, map size and resolution, integrator, track count, friction map or none, shared or per-rollout maps, a
given start state or the default one, flat / bumpy / off-map starts -- to catch indexing and masking slips."""
import numpy as np
import pytest
import torch

from tests import helpers as hp
from tests.test_rollout_gpu import make_dphysics

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _case(seed):
    from monoforce_amd import synthetic as syn
    rng = np.random.RandomState(1000 + seed)
    N = int(rng.choice([3, 4, 5, 9, 16, 17, 31, 40, 64, 65, 100, 129, 200]))
    n_tracks = int(rng.choice([2, 4]))
    B = int(rng.randint(1, 41))
    T = int(rng.randint(2, 61))
    H = int(rng.randint(10, 49))
    res = float(rng.choice([0.05, 0.1, 0.2]))
    d_max = H * res / 2
    integ = int(rng.randint(0, 2))
    shared = bool(rng.randint(0, 2))
    with_mu = bool(rng.randint(0, 3))
    with_state = bool(rng.randint(0, 2))
    pts, masks = syn.robot_points_box(N, seed=seed, n_tracks=n_tracks)
    nb = 1 if shared else B
    amp = float(rng.choice([0.0, 0.2, 0.6]))
    z = torch.stack([syn.bump_terrain(syn.bump_params(seed * 7 + b), d_max, res, torch.float64)[:H, :H] * amp for b in range(nb)])
    mu = torch.stack([syn.wave_friction(d_max, res, 0.4, 1.0, 1.0 + 0.3 * b, 0.7, torch.float64)[:H, :H] for b in range(nb)]) if with_mu else None
    ctrl = syn.varying_controls(B, T, seed=seed, dtype=torch.float64)
    state = None
    if with_state:
        from scipy.spatial.transform import Rotation
        f = lambda a: torch.as_tensor(a, dtype=torch.float64)  # noqa: E731
        span = d_max * float(rng.choice([0.3, 1.2]))       # 1.2: some rollouts start off the map (index clamping)
        state = (f(rng.uniform(-span, span, (B, 3)) * [1, 1, 0.05]), f(rng.uniform(-0.5, 0.5, (B, 3))),
                 f(Rotation.from_euler('xyz', rng.uniform(-0.2, 0.2, (B, 3)) * [1, 1, 15]).as_matrix()), f(rng.uniform(-0.3, 0.3, (B, 3))))
    return dict(N=N, n_tracks=n_tracks, B=B, T=T, H=H, res=res, d_max=d_max, integ=integ, shared=shared), pts, masks, z, mu, ctrl, state


@pytest.mark.parametrize('seed', range(24))
def test_random_shape_forward_and_gradients_vs_oracle_f64(seed):
    from oracle import dphysics_oracle as orc
    info, pts, masks, z, mu, ctrl, state = _case(seed)
    B, T = info['B'], info['T']
    spec = hp.spec_from(pts, masks, info['integ'], info['res'], info['d_max'])
    expand = lambda m: None if m is None else (m.expand(B, -1, -1) if info['shared'] else m)  # noqa: E731

    def run(fn, dev):
        zl = z.clone().to(dev).requires_grad_(True)
        cl = ctrl.clone().to(dev).requires_grad_(True)
        ml = None if mu is None else mu.clone().to(dev).requires_grad_(True)
        st = None
        if state is not None:
            st = [s.clone().to(dev) for s in state]
            for s in st[1:]:
                s.requires_grad_(True)
        outs = fn(expand(zl), cl, None if st is None else tuple(st), expand(ml))
        hp.probe_loss(outs, torch.float64).backward()
        grads = [zl.grad, cl.grad] + ([] if ml is None else [ml.grad]) + ([] if st is None else [s.grad for s in st[1:]])
        x0_after = None if st is None else st[0].detach().cpu()
        return [o.detach().cpu() for o in outs], [g.cpu() for g in grads], x0_after

    def f_oracle(zz, cc, st, mm):
        so, fo = orc.rollout(spec, zz, cc, state=st, friction=mm)
        return list(so) + list(fo)

    dp = make_dphysics(pts, masks, info['integ'], info['res'], info['d_max'])      # default time grid linspace(0, 5, 500)[:T]

    def f_hip(zz, cc, st, mm):
        s, f = dp(zz, cc, state=st, friction=mm)
        return list(s) + list(f)

    o_ref, g_ref, x_ref = run(f_oracle, 'cpu')
    o_hip, g_hip, x_hip = run(f_hip, DEV)
    for k, a, b in zip(hp.OUT_KEYS, o_hip, o_ref):
        assert a.shape == b.shape, (info, k, a.shape, b.shape)
        assert hp.rel_err(a, b) <= 1e-9, (info, k, hp.rel_err(a, b))
    for i, (a, b) in enumerate(zip(g_hip, g_ref)):
        assert hp.rel_err(a, b) <= 1e-7, (info, 'grad', i, hp.rel_err(a, b))
    if x_ref is not None:
        assert hp.rel_err(x_hip, x_ref) <= 1e-12       # the terrain snap written back into the caller's x0


@pytest.mark.parametrize('precise', [False, True])
@pytest.mark.parametrize('seed', range(12))
def test_random_shape_forward_f32_vs_oracle_f64(seed, precise):
    """float32 kernels (fast and exact arithmetic) on the same random shapes: <= 1e-4 relative on poses (north_star bar) at
    these horizons (T <= 60), 1e-3 on the force rows (they carry the contact model's amplification of pose rounding)."""
    from oracle import dphysics_oracle as orc
    info, pts, masks, z, mu, ctrl, state = _case(seed)
    B = info['B']
    spec = hp.spec_from(pts, masks, info['integ'], info['res'], info['d_max'])
    expand = lambda m: None if m is None else (m.expand(B, -1, -1) if info['shared'] else m)  # noqa: E731
    with torch.no_grad():
        so, fo = orc.rollout(spec, expand(z), ctrl, state=None if state is None else tuple(s.clone() for s in state), friction=expand(mu))
        dp = make_dphysics(pts, masks, info['integ'], info['res'], info['d_max'], precise=precise)
        f32 = lambda t: None if t is None else t.to(torch.float32).to(DEV)  # noqa: E731
        sh, fh = dp(expand(f32(z)), f32(ctrl), state=None if state is None else tuple(f32(s) for s in state), friction=expand(f32(mu)))
    for k, a, b in zip(hp.OUT_KEYS, list(sh) + list(fh), list(so) + list(fo)):
        tol = 1e-4 if k in ('Xs', 'Rs') else 1e-3
        assert hp.rel_err(a.cpu().double(), b) <= tol, (info, k, hp.rel_err(a.cpu().double(), b))
