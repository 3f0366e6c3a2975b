"""Shared test plumbing: golden-fixture loading and error metrics."""
import os

import numpy as np
import torch

from oracle import dphysics_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DT = {'f32': torch.float32, 'f64': torch.float64}
OUT_KEYS = ('Xs', 'Xds', 'Rs', 'Om', 'Fs', 'Ff')
SMALL = dict(grid_res=0.1, d_max=1.6, T=48)
FULL = dict(grid_res=0.05, d_max=6.4, T=500, B=4)


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def rel_err(a, b):
    """max |a-b| / max |b|  ("rel-to-absmax", the metric of SURVEY A.2)."""
    a, b = (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v for v in (a, b))
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def spec_from(points, masks, integ, grid_res, d_max, mass=40.0):
    pts32 = torch.as_tensor(points, dtype=torch.float32)
    Ly = float(pts32[:, 1].max() - pts32[:, 1].min())          # robot_size[1] (dphys_config.py:43,68), float32 value
    return orc.RolloutSpec(points=pts32, driving_parts=[torch.as_tensor(m) for m in masks], robot_size_y=Ly, mass=mass,
                           grid_res=grid_res, d_max=d_max, integrator=integ)


def small_case(g, name, dtype):
    """Inputs of small case `name` from rollout_small.npz as torch tensors of `dtype`."""
    t = lambda k: torch.as_tensor(g[f'{name}/{k}']).to(dtype)  # noqa: E731
    z, ctrl = t('z'), t('ctrl')
    mu = t('mu') if f'{name}/mu' in g.files else None
    state = tuple(t(k) for k in ('x0', 'xd0', 'R0', 'w0')) if f'{name}/x0' in g.files else None
    return g[f'{name}/points'], g[f'{name}/masks'], z, ctrl, state, mu


def full_inputs(dtype):
    """Regenerate the full-horizon inputs from their parameters (same code as gen_golden.full_inputs)."""
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    pr = [syn.bump_params(11), syn.bump_params(12), syn.bump_params(0, smooth=True), np.array([[0.0, 0.0, 0.0, 1.0]])]
    z = torch.stack([syn.bump_terrain(p, FULL['d_max'], FULL['grid_res'], torch.float64) for p in pr]).to(dtype)
    mu = torch.stack([syn.wave_friction(FULL['d_max'], FULL['grid_res'], 0.5, 1.0, 1.3 + 0.1 * k, 0.9, torch.float64)
                      for k in range(FULL['B'])]).to(dtype)
    ctrl = syn.const_controls(FULL['B'], FULL['T'], seed=7, dtype=torch.float64).to(dtype)
    return pts, masks, z, mu, ctrl


def probe_loss(outs, dtype):
    from monoforce_amd import synthetic as syn
    scales = [1.0, 1.0, 1.0, 1.0, 1e-3, 1e-3]
    loss = 0
    for i, (o, s) in enumerate(zip(outs, scales)):
        loss = loss + (o * syn.probe_weights(o.shape, phase=0.5 + i, dtype=dtype).to(o.device)).sum() * s
    return loss
