"""SURVEY 8 row a5 on its own entry point: `DPhysics.interpolate_grid` (mf_interpolate_grid_*, the rollout kernels' device code)
against the reference's golden vectors and -- the index half, as integers -- against `((q + d_max) / grid_res).long()` itself.
VERDICT r2 item 2a: the fast-math kernels used to scale by 1 / res; now the cell coordinate is the correctly rounded quotient in
both arithmetic modes, so every query, on a cell edge or not, lands in the reference's cell."""
import numpy as np
import pytest
import torch

from tests import helpers as hp

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _dp(res, d_max, precise):
    from monoforce_amd import synthetic as syn
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import DPhysics
    pts, masks = syn.robot_points_4()
    cfg = DPhysConfig(robot='tradr', grid_res=res, robot_points=pts, driving_parts=masks)
    cfg.d_max = d_max
    return DPhysics(cfg, device=DEV, precise=precise)


@pytest.mark.parametrize('tag', ['f32', 'f64'])
@pytest.mark.parametrize('precise', [False, True])
def test_golden_values_and_normals(tag, precise):
    """All 16 edge / out-of-range queries of tests/golden/interp.npz (outputs of the reference's own interpolate_grid), both grids,
    in BOTH arithmetic modes -- no query is excused for sitting on a cell boundary any more."""
    g = hp.load('interp')
    dp = _dp(float(g['grid_res']), float(g['d_max']), precise)
    grid, qx, qy = (torch.as_tensor(g[f'{tag}/{k}']) for k in ('grid', 'qx', 'qy'))
    z, n = dp.interpolate_grid(grid, qx, qy, return_normals=True)
    zr, nr = g[f'{tag}/z'].astype(np.float64), g[f'{tag}/n'].astype(np.float64)
    tol = 1e-12 if tag == 'f64' else 2e-6
    assert np.abs(z.cpu().numpy() - zr).max() <= tol * np.abs(zr).max()
    assert np.abs(n.cpu().numpy() - nr).max() <= (1e-10 if tag == 'f64' else 2e-6)
    if tag == 'f32' and precise:          # the exact mode is the reference's op sequence: the same bits
        assert np.array_equal(z.cpu().numpy(), g['f32/z'])


def _torch_index(q, d_max, res, H, W):
    """The reference's index arithmetic on the CPU, float32 (dphysics.py:419-435)."""
    xi = ((q[0] + d_max) / res).long()
    yi = ((q[1] + d_max) / res).long()
    fx = (q[0] + d_max) / res - xi.float()
    fy = (q[1] + d_max) / res - yi.float()
    cells = torch.stack([yi + H * xi, yi + H * (xi + 1), (yi + 1) + H * xi, (yi + 1) + H * (xi + 1)], -1).clamp(0, H * W - 1)
    return cells, torch.stack([fx, fy], -1)


@pytest.mark.parametrize('precise', [False, True])
@pytest.mark.parametrize('res,d_max', [(0.05, 6.4), (0.1, 6.4), (0.1, 1.6), (0.07, 3.0)])
def test_cell_index_is_the_references_integer(precise, res, d_max):
    """10^6 float32 positions -- uniformly random, exactly ON cell edges, one ulp either side of them, and off the map -- : the
    clamped flat cell indices and the fractions of the HIP kernels equal torch's `((q + d_max) / grid_res).long()` arithmetic
    bit for bit (integers; the fractions as float32 bit patterns), in fast and in exact mode."""
    H = W = int(round(2 * d_max / res))
    g = torch.Generator().manual_seed(1)
    n = 250_000
    rnd = (torch.rand(2, n, generator=g) * 2.4 - 1.2) * d_max                      # incl. 20 % beyond the map on either side
    k = torch.randint(-8, H + 8, (2, n), generator=g).float()
    edge = (k * np.float32(res)).float() - np.float32(d_max)                     # on a cell edge (as float32 arithmetic lands it)
    up = torch.nextafter(edge, torch.full_like(edge, 1e9))
    dn = torch.nextafter(edge, torch.full_like(edge, -1e9))
    q = torch.cat([rnd, edge, up, dn], 1).float()                                # [2, 10^6]
    dp = _dp(res, d_max, precise)
    grid = torch.arange(H * W, dtype=torch.float32).view(1, H, W)
    z, cells, frac = dp.interpolate_grid(grid, q[0:1], q[1:2], return_cells=True)
    ref_cells, ref_frac = _torch_index(q, np.float32(d_max).item(), res, H, W)
    assert torch.equal(cells[0].cpu().long(), ref_cells)
    assert torch.equal(frac[0].cpu().view(torch.int32), ref_frac.view(torch.int32))


def test_shapes_and_refusals():
    dp = _dp(0.1, 1.6, False)
    grid = torch.zeros(3, 32, 32)
    z = dp.interpolate_grid(grid, torch.zeros(3, 5), torch.zeros(3, 5))
    assert z.shape == (3, 5) and z.is_cuda
    with pytest.raises(RuntimeError):
        dp.interpolate_grid(grid.requires_grad_(True), torch.zeros(3, 5), torch.zeros(3, 5))
    with pytest.raises(TypeError):
        dp.interpolate_grid(grid.detach().half(), torch.zeros(3, 5), torch.zeros(3, 5))
