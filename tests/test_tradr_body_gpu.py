"""The HIP rollout on the reference's REAL 175-point tradr body (config/meshes/tradr.obj through the reference's own
`robot_geometry`; fixture tests/golden/tradr_body.npz, see gen_golden.py::gen_tradr) -- every other large-body test runs a synthetic
box.  Outputs and gradients against the REFERENCE's own rollouts of that body (float64 at 1e-9 / 1e-7; float32 fast math at the
north_star bar), for the general kernels, the recording forward + record-reading backward (the default route at this size) and the
float64 validation build of the latter; and the bench's n175 launch shape (64 rollouts, shared map) against the float64 oracle."""
import numpy as np
import pytest
import torch

from tests import helpers as hp
from tests.test_rollout_gpu import make_dphysics

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _fixture(dt):
    g = hp.load('tradr_body')
    d_max, res, _ = (float(v) for v in g['meta'])
    z, mu, ctrl = (torch.as_tensor(g[k]).to(dt) for k in ('z', 'mu', 'ctrl'))
    return g, g['points'], list(g['masks']), z, mu, ctrl, d_max, res


def _run(dp, z, mu, ctrl, dt):
    zl, ml, cl = (t.clone().to(DEV).requires_grad_(True) for t in (z, mu, ctrl))
    states, forces = dp(zl, cl, friction=ml)
    outs = list(states) + list(forces)
    hp.probe_loss(outs, dt).backward()
    torch.cuda.synchronize()
    return [o.detach().cpu() for o in outs], [t.grad.cpu() for t in (zl, ml, cl)]


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('ppl', [0, 4, 16])      # 0: the library's choice; 4: several points per lane; 16: MF_LANES_COMPONENT = the validation build
def test_real_tradr_body_f64_vs_reference(integ, ppl):
    g, pts, masks, z, mu, ctrl, d_max, res = _fixture(torch.float64)
    dp = make_dphysics(pts, masks, integ, res, d_max, points_per_lane=ppl)
    outs, grads = _run(dp, z, mu, ctrl, torch.float64)
    for k, o in zip(hp.OUT_KEYS, outs):
        assert hp.rel_err(o, g[f'f64/i{integ}/{k}']) <= 1e-9, (k, hp.rel_err(o, g[f'f64/i{integ}/{k}']))
    for k, a in zip(('g_z', 'g_mu', 'g_ctrl'), grads):
        assert hp.rel_err(a, g[f'f64/i{integ}/{k}']) <= 1e-7, (k, hp.rel_err(a, g[f'f64/i{integ}/{k}']))


@pytest.mark.parametrize('integ', [0, 1])
@pytest.mark.parametrize('ppl', [0, 4])
def test_real_tradr_body_f32_vs_reference(integ, ppl):
    """float32 fast math (ppl = 0: recording forward over 4 waves + record-reading backward) against the reference's own FLOAT32
    rollout of the same float32 inputs: poses within north_star's 1e-4 over these 48 steps.  (The fixture's float64
    rollout started from the float64 inputs: its distance to ANY float32 run is the input rounding, 1.0e-3 on Xs -- not a kernel
    property.)  Gradients against the float64 oracle on the float32-valued inputs, at the bar derived from the oracle alone:
    max(2e-3, 3 x the distance between ITS float32 and float64 gradients)."""
    from oracle import dphysics_oracle as orc      # checker only
    g, pts, masks, z, mu, ctrl, d_max, res = _fixture(torch.float32)
    dp = make_dphysics(pts, masks, integ, res, d_max, points_per_lane=ppl)
    outs, grads = _run(dp, z, mu, ctrl, torch.float32)
    for k, o in zip(hp.OUT_KEYS, outs):
        key = f'f32/i{integ}/{k}'
        if key in g.files:
            if k in ('Xs', 'Rs'):      # (velocities and forces: the calm-prefix protocol below)
                assert hp.rel_err(o, g[key]) <= 1e-4, (k, hp.rel_err(o, g[key]))
    spec = hp.spec_from(pts, masks, integ, res, d_max)

    def oracle_grads(dt):
        zo, mo, co = (t.clone().to(dt).requires_grad_(True) for t in (z, mu, ctrl))
        so, fo = orc.rollout(spec, zo, co, friction=mo)
        hp.probe_loss(list(so) + list(fo), dt).backward()
        return [zo.grad, mo.grad, co.grad], list(so) + list(fo)
    g64, o64 = oracle_grads(torch.float64)
    g32, o32 = oracle_grads(torch.float32)
    # north_star's <= 1e-4 on poses AND forces, by the calm-prefix protocol of test_rollout_gpu.py (the flat 1e-3 / 2e-3 bars were
    # argued, not shown per step): per rollout and step, every output -- velocities and both force tensors included -- is within 1e-4 of the
    # float64 oracle (on the same float32-valued inputs) for as long as the oracle's OWN float32 run has stayed within 1e-6 of its float64 run
    # (running maximum: once a contact flips in float32 the rollout has left the part float32 can referee); beyond, boundedness
    n_calm, n_all, worst = 0, 0, 0.0
    for k, o, b, e32 in zip(hp.OUT_KEYS, outs, o64, o32):
        o_, b_, e_ = (t.detach().cpu().double().numpy() for t in (o, b, e32))
        B, T = b_.shape[:2]
        scale = np.abs(b_).reshape(B, -1).max(1).clip(1e-30)[:, None]
        env = np.maximum.accumulate(np.abs(e_ - b_).reshape(B, T, -1).max(2) / scale, axis=1)
        err = np.abs(o_ - b_).reshape(B, T, -1).max(2) / scale
        calm = env <= 1e-6
        n_calm += int(calm.sum()); n_all += calm.size
        assert (err[calm] <= 1e-4).all(), (k, float(err[calm].max()), int(calm.sum()), calm.size)
        assert np.isfinite(o_).all() and float(err.max()) <= max(0.5, 3.0 * float((np.abs(e_ - b_).reshape(B, T, -1).max(2) / scale).max())), (k, float(err.max()))
        # ... and beyond the calm prefix the error stays a bounded multiple of the oracle's own float32 drift on that rollout and step
        ratio = err / np.maximum(env, 1e-6)
        worst = max(worst, float(ratio[~calm].max()) if (~calm).any() else 0.0)
    assert n_calm >= 36, (n_calm, n_all)      # (not vacuous: with 175 contact points float32 leaves the 1e-6 envelope within a few steps)
    assert worst <= 100.0, worst
    for k, a, b64, b32 in zip(('g_z', 'g_mu', 'g_ctrl'), grads, g64, g32):
        bar = max(2e-3, 3.0 * hp.rel_err(b32, b64))
        assert torch.isfinite(a).all(), k
        assert hp.rel_err(a, b64) <= bar, (k, hp.rel_err(a, b64), 'bar', bar)


@pytest.mark.parametrize('integ', [1, 0])
def test_n175_launch_shape_f64_build_vs_oracle(integ):
    """The bench's n175 shape in small: 64 rollouts of the real body on ONE shared map pair (the launch shape that picks one rollout per
    workgroup of four waves and the LDS gradient tile), 40 steps, float64 validation build against the float64 oracle."""
    from monoforce_amd import synthetic as syn
    from oracle import dphysics_oracle as orc      # checker only
    g = hp.load('tradr_body')
    pts, masks = g['points'], list(g['masks'])
    B, T, d_max, res = 64, 40, 6.4, 0.1
    z = (syn.bump_terrain(syn.bump_params(3), d_max, res, torch.float64) * 0.5).unsqueeze(0)
    mu = syn.wave_friction(d_max, res, 0.5, 1.0, 1.1, 0.7, torch.float64).unsqueeze(0)
    ctrl = syn.varying_controls(B, T, seed=5, dtype=torch.float64)
    spec = hp.spec_from(pts, masks, integ, res, d_max)
    zo, mo, co = (t.clone().requires_grad_(True) for t in (z, mu, ctrl))
    so, fo = orc.rollout(spec, zo.expand(B, -1, -1), co, friction=mo.expand(B, -1, -1))
    Xo = so[0]
    w = syn.probe_weights(Xo[:, ::5].shape, phase=0.3, dtype=torch.float64)
    (Xo[:, ::5] * w).sum().backward()
    dp = make_dphysics(pts, masks, integ, res, d_max, points_per_lane=16)
    zl, ml, cl = (t.clone().to(DEV).requires_grad_(True) for t in (z, mu, ctrl))
    sh, fh = dp(zl, cl, friction=ml)
    (sh[0][:, ::5] * w.to(DEV)).sum().backward()
    for k, a, b in zip(hp.OUT_KEYS, list(sh) + list(fh), list(so) + list(fo)):
        assert hp.rel_err(a.detach().cpu(), b.detach()) <= 1e-9, (k, hp.rel_err(a.detach().cpu(), b.detach()))
    for k, a, b in (('z', zl.grad, zo.grad), ('mu', ml.grad, mo.grad), ('controls', cl.grad, co.grad)):
        assert hp.rel_err(a.cpu(), b) <= 1e-7, (k, hp.rel_err(a.cpu(), b))
