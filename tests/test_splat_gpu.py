"""Parity of the HIP BEV splat (through the C ABI) against the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import splat_oracle as so
from tests import helpers as hp

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def pool(geom, x, dx, bx, nx, requires_grad=False):
    from monoforce_amd.splat import voxel_pooling
    xg = torch.as_tensor(x).to(DEV).requires_grad_(requires_grad)
    out = voxel_pooling(torch.as_tensor(geom).to(DEV), xg, torch.as_tensor(dx), torch.as_tensor(bx), torch.as_tensor(nx))
    return out, xg


@pytest.mark.parametrize('tag', ['f32', 'f64'])
def test_golden_forward_backward(tag):
    g = hp.load('lss')
    x = g['x'].astype(np.float32 if tag == 'f32' else np.float64)
    out, xg = pool(g['geom'], x, g['dx'], g['bx'], g['nx'], requires_grad=True)
    assert tuple(out.shape) == g['pooled_exact_f64'].shape
    # vs the exact (float64-accumulated) sums: float32 accumulation of <= ~30 terms per voxel
    assert hp.rel_err(out, g['pooled_exact_f64']) <= (3e-7 if tag == 'f32' else 1e-15)
    # and as close to the reference's own float32 output as that output is to the truth
    assert hp.rel_err(out, g['pooled_ref_f32']) <= 1e-5
    from monoforce_amd import synthetic as syn
    w = syn.probe_weights(out.shape, phase=0.3, dtype=out.dtype).to(DEV)
    (out * w).sum().backward()
    assert np.array_equal(xg.grad.cpu().numpy().astype(np.float32), g['g_x'])     # pure gather: bit-exact with QuickCumsum.backward
    out2, _ = pool(g['geom'], x, g['dx'], g['bx'], g['nx'])
    assert torch.equal(out, out2), 'forward must be bit-reproducible (sorted per-voxel summation order)'


def _c4_problem(B=1, C=64, seed=0, n=256, res=0.05):
    """BASELINE config-4 shapes: 4 cams x 3x256x512 images /16 -> fH=16, fW=32, D=59; 256x256 BEV at 0.05 m."""
    from monoforce_amd import synthetic as syn
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    grid_conf = dict(xbound=[-6.4, 6.4, res], ybound=[-6.4, 6.4, res], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
    m = LiftSplatShoot(grid_conf, dict(final_dim=(256, 512)), outC=1, build_backbones=False)
    rig = syn.lss_camera_rig(B, 4, 256, 512, 300.0)
    geom = m.get_geometry(*rig)
    rng = np.random.RandomState(seed)
    x = rng.randn(*geom.shape[:-1], C).astype(np.float32)
    return m, geom, x


def test_full_size_vs_oracle():
    m, geom, x = _c4_problem()
    assert geom.shape == (1, 4, 59, 16, 32, 3)
    out, xg = pool(geom, x, m.dx, m.bx, m.nx, requires_grad=True)
    ref, kept = so.voxel_pooling(geom.numpy(), x, m.dx.numpy(), m.bx.numpy(), m.nx.numpy())
    assert 0.2 < kept.mean() <= 1.0
    assert hp.rel_err(out, ref) <= 5e-7
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
    out.backward(gout.to(DEV))
    gref = so.voxel_pooling_grad(geom.numpy(), gout.numpy(), m.dx.numpy(), m.bx.numpy(), m.nx.numpy(), C=64)
    assert np.array_equal(xg.grad.cpu().numpy().reshape(gref.shape), gref)


def test_size_independent_properties():
    m, geom, x = _c4_problem(B=2, C=64, seed=3)
    xd = x.astype(np.float64)
    out, _ = pool(geom, xd, m.dx, m.bx, m.nx)
    _, kept = so.voxel_pooling(geom.numpy(), x[..., :1], m.dx.numpy(), m.bx.numpy(), m.nx.numpy())
    # conservation: everything that was kept ends up in the grid, nothing else does
    assert abs(float(out.sum()) - xd.reshape(2, -1, 64)[kept].sum()) <= 1e-8 * np.abs(xd).sum()
    # linearity
    y = np.random.RandomState(9).randn(*x.shape)
    a, _ = pool(geom, 2.0 * xd - 3.0 * y, m.dx, m.bx, m.nx)
    b, _ = pool(geom, y, m.dx, m.bx, m.nx)
    assert hp.rel_err(a, (2.0 * out - 3.0 * b).cpu()) <= 1e-12
    # batch independence: sample 1 alone gives the same slab
    o1, _ = pool(geom[1:], xd[1:], m.dx, m.bx, m.nx)
    assert torch.equal(o1[0], out[1])
    # zero features -> zero grid; every cell is written (no stale memory)
    z, _ = pool(geom, np.zeros_like(x), m.dx, m.bx, m.nx)
    assert float(z.abs().max()) == 0.0


@pytest.mark.parametrize('C,nxy,nz', [(8, 16, 1), (80, 50, 1), (64, 40, 3), (1, 7, 2), (130, 12, 1)])
def test_ragged_shapes_and_edges(C, nxy, nz):
    """Channel counts that are not multiples of 64, planes that are not multiples of the 64-voxel tile, nz > 1,
    NaN / inf / far-away / exactly-on-the-boundary coordinates, all-dropped and single-voxel pile-ups (> 64 points)."""
    rng = np.random.RandomState(C + nxy)
    B, P = 2, 700
    dx = np.array([0.5, 0.5, 1.0], np.float32)
    bx = np.array([-nxy * 0.25 + 0.25, -nxy * 0.25 + 0.25, -nz * 0.5 + 0.5], np.float32)
    nx = np.array([nxy, nxy, nz])
    geom = (rng.rand(B, P, 3).astype(np.float32) - 0.5) * np.array([nxy * 0.6, nxy * 0.6, nz * 1.2], np.float32)
    geom[0, :100] = np.array([0.1, 0.1, 0.0], np.float32)                 # 100 points piled into one voxel
    geom[0, 100] = [np.nan, 0, 0]; geom[0, 101] = [np.inf, 0, 0]; geom[0, 102] = [0, -np.inf, 0]; geom[0, 103] = [1e30, 0, 0]
    geom[0, 104] = [-nxy * 0.25, 0, 0]; geom[0, 105] = [-nxy * 0.25 - 0.49, 0, 0]   # on / just below the lower bound: kept (trunc)
    geom[1, :] += 1000.0                                                  # sample 1: everything dropped
    x = rng.randn(B, P, C)
    out, xg = pool(geom, x, dx, bx, nx, requires_grad=True)
    ref, kept = so.voxel_pooling(geom, x, dx, bx, nx)
    assert kept[0, 104] and kept[0, 105] and not kept[0, 100:104].any() and not kept[1].any()
    assert hp.rel_err(out, ref) <= 1e-13
    assert float(out[1].abs().max()) == 0.0
    gout = rng.randn(*out.shape)
    out.backward(torch.as_tensor(gout).to(DEV))
    assert np.array_equal(xg.grad.cpu().numpy(), so.voxel_pooling_grad(geom, gout, dx, bx, nx, C))


def test_plan_reuse():
    from monoforce_amd.splat import SplatPlan, voxel_pooling
    m, geom, x = _c4_problem(seed=5)
    plan = SplatPlan(geom.to(DEV), m.dx, m.bx, m.nx)
    a = voxel_pooling(None, torch.as_tensor(x).to(DEV), None, None, None, plan=plan)
    b = voxel_pooling(geom.to(DEV), torch.as_tensor(x).to(DEV), m.dx, m.bx, m.nx)
    c = voxel_pooling(None, torch.as_tensor(x[..., :16].copy()).to(DEV), None, None, None, plan=plan)   # other C, same plan
    assert torch.equal(a, b) and torch.equal(c, a[:, :16])


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('shape', [dict(B=1, N=4, D=59, fH=16, fW=32, C=64, nx=256, nz=1), dict(B=2, N=3, D=7, fH=5, fW=9, C=80, nx=40, nz=2),
                                   dict(B=1, N=1, D=70, fH=3, fW=4, C=5, nx=24, nz=1),
                                   # (round 4: the fused forward turns a point id into (camera, pixel) by multiply-high -- odd divisors, 1, a prime)
                                   dict(B=2, N=2, D=59, fH=7, fW=11, C=64, nx=48, nz=1), dict(B=1, N=5, D=3, fH=1, fW=1, C=8, nx=16, nz=1),
                                   dict(B=1, N=2, D=13, fH=1, fW=97, C=12, nx=32, nz=2)])
def test_fused_lift_splat_equals_lift_then_splat(dtype, shape):
    """mf_bev_lift_splat_* (depth x context inside the splat) vs the materialised lift followed by the plain splat: forward,
    and the gradients w.r.t. depth and context, incl. C not a multiple of 64, D > 64, nz > 1 and dropped points."""
    from monoforce_amd import splat
    B, N, D, fH, fW, C, nx, nz = (shape[k] for k in ('B', 'N', 'D', 'fH', 'fW', 'C', 'nx', 'nz'))
    g = torch.Generator().manual_seed(5)
    half = nx * 0.05 / 2
    geom = (torch.rand(B, N, D, fH, fW, 3, generator=g) * 2 - 1) * torch.tensor([half * 1.3, half * 1.3, 3.0])   # ~40 % outside in x or y
    dx = torch.tensor([0.05, 0.05, 6.4 / nz]); bx = torch.tensor([-half + 0.025, -half + 0.025, -3.2 + 3.2 / nz]); nxv = torch.tensor([nx, nx, nz])
    depth0 = torch.rand(B * N, D, fH, fW, generator=g).softmax(dim=1).to(dtype)
    ctx0 = torch.randn(B * N, C, fH, fW, generator=g).to(dtype)
    w = torch.randn(B, C * nz, nx, nx, generator=g).to(dtype).to(DEV)
    plan = splat.SplatPlan(geom.to(DEV), dx, bx, nxv)

    d1, c1 = depth0.to(DEV).requires_grad_(True), ctx0.to(DEV).requires_grad_(True)
    x = (d1.unsqueeze(1) * c1.unsqueeze(2)).view(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2)          # lss.py:63-71, 226-236
    ref = splat.voxel_pooling(geom.to(DEV), x, dx, bx, nxv, plan=plan)
    (ref * w).sum().backward()

    d2, c2 = depth0.to(DEV).requires_grad_(True), ctx0.to(DEV).requires_grad_(True)
    out = splat.lift_voxel_pooling(geom.to(DEV), d2, c2, dx, bx, nxv, plan=plan)
    (out * w).sum().backward()
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert torch.equal(out, ref)                       # same products (rounded before the add), same summation order
    assert hp.rel_err(d2.grad.cpu(), d1.grad.cpu()) <= tol, hp.rel_err(d2.grad.cpu(), d1.grad.cpu())
    assert hp.rel_err(c2.grad.cpu(), c1.grad.cpu()) <= tol, hp.rel_err(c2.grad.cpu(), c1.grad.cpu())


def _voxel_keys(idx, kept, B, n_per_sample, nx, ny, nz):
    b = np.arange(idx.shape[0]) // n_per_sample
    key = ((b * nz + idx[:, 2]) * nx + idx[:, 0]) * ny + idx[:, 1]
    return np.where(kept, key, -1)


def _face_distance(geom, off, dx):
    """Distance (in voxels) of every point to the nearest voxel face, minimum over the three axes, in float64."""
    u = (geom.double().view(-1, 3) - torch.as_tensor(off, dtype=torch.float64, device=geom.device)) / torch.as_tensor(dx, dtype=torch.float64, device=geom.device)
    return (u - torch.round(u)).abs().min(dim=1).values


@pytest.mark.parametrize('host_inverse', [True, False])
def test_camera_plan_reproduces_reference_indices(host_inverse):
    """get_geometry fused into the key pass, pinned on the reference itself: the golden rig (with image augmentation) through
    `SplatPlan.from_cameras` must give the voxel of every frustum point that the reference's get_geometry + voxel_pooling
    index arithmetic gave (tests/golden/lss.npz: voxel_idx, kept).  The generator overwrote the geometry of points 0..3 with
    hand-placed edge cases (gen_golden.py, fixture 5), so those four are not functions of the rig and are skipped.
    host_inverse=True: torch.inverse on the device (round 4's route), every index equal.  False (the default route, round 5): the
    inversions inside the key kernel (float64 adjugate rounded once) -- the golden rig's augmentation has an in-plane rotation, so the
    inverse's last bits may differ from LAPACK's: a point may change voxel only if it lies on a voxel face to within 1e-4 voxel, and
    at most 2 in 10 000 do."""
    from monoforce_amd import splat
    g = hp.load('lss')
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    plan = splat.SplatPlan.from_cameras(t('frustum'), t('rots'), t('trans'), t('intrins'), t('post_rots'), t('post_trans'),
                                        torch.from_numpy(g['dx']), torch.from_numpy(g['bx']), torch.from_numpy(g['nx']), host_inverse=host_inverse)
    B = g['rots'].shape[0]
    nx, ny, nz = (int(v) for v in g['nx'])
    want = _voxel_keys(g['voxel_idx'], g['kept'], B, plan.n_per_sample, nx, ny, nz)
    assert 0.2 < g['kept'].mean() < 1.0
    got = plan.keys().cpu().numpy()
    if host_inverse:
        np.testing.assert_array_equal(got[4:], want[4:])
        return
    diff = got[4:] != want[4:]
    assert diff.mean() <= 2e-4, diff.mean()
    if diff.any():
        off = (torch.from_numpy(g['bx']) - torch.from_numpy(g['dx']) / 2.).tolist()
        d = _face_distance(torch.from_numpy(g['geom']).reshape(-1, 3)[4:][torch.from_numpy(diff)], off, g['dx'].tolist())
        assert float(d.max()) <= 1e-4, float(d.max())


@pytest.mark.parametrize('shape', [dict(B=1, N=4, H=256, W=512, bound=6.4, res=0.05), dict(B=3, N=5, H=96, W=160, bound=3.2, res=0.1),
                                   dict(B=2, N=1, H=64, W=96, bound=1.6, res=0.2)])
@pytest.mark.parametrize('host_inverse,rotate', [(True, True), (False, False), (False, True)])
def test_camera_plan_equals_geometry_plan(shape, host_inverse, rotate):
    """Random augmented rigs: keys from the camera models == keys from the materialised get_geometry tensor, bit for bit, and
    so is everything built from them (CSR lists -> pooled output) -- with torch's own inverses (host_inverse=True), and with the
    inversions inside the key kernel for the augmentations whose inverse is exact to one rounding (resize + crop + flip:
    rotate=False).  With an in-plane rotation the in-kernel inverse may differ from the LU's in the last bit: a point may change voxel
    only if it lies within 1e-4 voxel of a face, at most 2 in 10 000 do, and the pooled output is the exact sum over the plan's own keys."""
    from monoforce_amd import splat, synthetic as syn
    from monoforce_amd.terrain_encoder import LiftSplatShoot
    B, N, H, W, bound, res = (shape[k] for k in ('B', 'N', 'H', 'W', 'bound', 'res'))
    gc = dict(xbound=[-bound, bound, res], ybound=[-bound, bound, res], zbound=[-1.0, 1.0, 1.0], dbound=[0.6, bound, 0.1])
    m = LiftSplatShoot(gc, dict(final_dim=(H, W)), build_backbones=False).to(DEV)
    gen = torch.Generator().manual_seed(11)
    rots, trans, intrins, post_rots, post_trans = syn.lss_camera_rig(B, N, H, W, 0.6 * W)
    # image-space augmentation as the reference's datasets produce it (resize + crop + flip + small rotation, utils.py:52-76)
    for b in range(B):
        for n in range(N):
            s = 0.8 + 0.4 * float(torch.rand(1, generator=gen))
            a = 0.2 * (float(torch.rand(1, generator=gen)) - 0.5) * (1.0 if rotate else 0.0)
            flip = -1.0 if float(torch.rand(1, generator=gen)) < 0.5 else 1.0
            A = torch.tensor([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]], dtype=torch.float32) * s
            A[:, 0] *= flip
            post_rots[b, n, :2, :2] = A
            post_trans[b, n, :2] = (torch.rand(2, generator=gen) - 0.5) * 40
            trans[b, n] += (torch.rand(3, generator=gen) - 0.5) * 0.2
    rig = [t.to(DEV) for t in (rots, trans, intrins, post_rots, post_trans)]
    geom = m.get_geometry(*rig)
    p_geom = splat.SplatPlan(geom, m.dx, m.bx, m.nx)
    p_cam = splat.SplatPlan.from_cameras(m.frustum, *rig, m.dx, m.bx, m.nx, host_inverse=host_inverse)
    k1, k2 = p_geom.keys(), p_cam.keys()
    assert 0.05 < float((k1 >= 0).float().mean()) < 0.98
    x = torch.randn(B * p_cam.n_per_sample, 8, generator=gen).to(DEV).view(B, -1, 8)
    out_cam = splat.voxel_pooling(None, x, m.dx, m.bx, m.nx, plan=p_cam)
    if host_inverse or not rotate:
        assert torch.equal(k1, k2)
        assert torch.equal(out_cam, splat.voxel_pooling(geom, x, m.dx, m.bx, m.nx, plan=p_geom))
        return
    diff = k1 != k2
    assert float(diff.float().mean()) <= 2e-4, float(diff.float().mean())
    if bool(diff.any()):
        off, dxl, _ = splat.grid_host(m.dx, m.bx, m.nx)
        assert float(_face_distance(geom.reshape(-1, 3)[diff], off, dxl).max()) <= 1e-4
    # the pooled output is the exact sum over the plan's OWN keys
    nx, ny, nz = (int(v) for v in m.nx)
    kept = k2 >= 0
    ref = torch.zeros(B * nz * nx * ny, 8, dtype=torch.float64, device=DEV).index_add_(0, k2[kept].long(), x.reshape(-1, 8)[kept].double())
    ref = ref.view(B, nz, nx, ny, 8).permute(0, 1, 4, 2, 3).reshape(B, nz * 8, nx, ny)
    assert hp.rel_err(out_cam.cpu(), ref.cpu()) <= 3e-7


@pytest.mark.parametrize('B', [1, 3])
def test_plan_pass_v2_equals_the_round4_pass(B):
    """The four-launch plan pass (histogram atomics that return the arrival slot, single-pass look-back scan, atomic-free CSR fill) builds
    the SAME workspace arrays as round 4's seven-launch pass (MF_SPLAT_PLAN=0, run in a subprocess: the library reads the switch once):
    keys, CSR offsets and the per-voxel lists in ascending point order -- bit for bit, incl. dropped points and empty voxels."""
    import os, subprocess, sys, tempfile
    code = (
        "import sys, torch, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from monoforce_amd import splat, synthetic as syn\n"
        "from monoforce_amd.terrain_encoder import LiftSplatShoot\n"
        "B = %d\n"
        "gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])\n"
        "m = LiftSplatShoot(gc, dict(final_dim=(256, 512)), build_backbones=False).cuda()\n"
        "rig = [t.cuda() for t in syn.lss_camera_rig(B, 4, 256, 512, 300.0)]\n"
        "geom = m.get_geometry(*rig)\n"
        "geom.view(-1, 3)[::97] = float('nan')\n"
        "plans = [splat.SplatPlan(geom, m.dx, m.bx, m.nx), splat.SplatPlan.from_cameras(m.frustum, *rig, m.dx, m.bx, m.nx, host_inverse=True)]\n"
        "torch.cuda.synchronize()\n"
        "P = plans[0].B * plans[0].n_per_sample; V = plans[0].B * plans[0].nz * plans[0].nx * plans[0].ny\n"
        "a256 = lambda n: (n * 4 + 255) // 256 * 256\n"
        "out = {}\n"
        "for i, p in enumerate(plans):\n"
        "    w = p.workspace.cpu().numpy()\n"
        "    o_keys, o_off, o_list = 0, a256(P) + 2 * a256(V), a256(P) + 2 * a256(V) + a256(V + 1)\n"
        "    out['keys%%d' %% i] = w[o_keys:o_keys + 4 * P].view(np.int32)\n"
        "    out['off%%d' %% i] = w[o_off:o_off + 4 * (V + 1)].view(np.int32)\n"
        "    n = int(out['off%%d' %% i][-1])\n"
        "    out['list%%d' %% i] = w[o_list:o_list + 4 * n].view(np.int32)\n"
        "np.savez(sys.argv[1], **out)\n") % (hp.REPO if hasattr(hp, 'REPO') else os.path.dirname(os.path.dirname(os.path.abspath(__file__))), B)
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, env in (('v2', {}), ('v1', {'MF_SPLAT_PLAN': '0'})):
            path = os.path.join(td, tag + '.npz')
            r = subprocess.run([sys.executable, '-c', code, path], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            res[tag] = dict(np.load(path))
    assert set(res['v1']) == set(res['v2']) and len(res['v1']) == 6
    for k in res['v1']:
        assert res['v1'][k].shape == res['v2'][k].shape and np.array_equal(res['v1'][k], res['v2'][k]), k
    assert int(res['v2']['off0'][-1]) > 1000 and (res['v2']['keys0'] < 0).any()


def _random_pool_case(seed):
    """Random pooling problem with points clustered into few voxels (long per-wave point lists, voxel changes anywhere inside the
    64-point chunks, chunk counts on and off multiples of 64)."""
    rng = np.random.RandomState(9000 + seed)
    B = int(rng.randint(1, 4))
    C = int(rng.choice([1, 3, 16, 63, 64, 65, 100, 130]))
    nxy, nz = int(rng.randint(5, 71)), int(rng.randint(1, 4))
    P = int(rng.choice([1, 63, 64, 65, 128, 500, 1500, 4000]))
    dx = np.array([0.5, 0.5, 1.0], np.float32)
    bx = np.array([-nxy * 0.25 + 0.25, -nxy * 0.25 + 0.25, -nz * 0.5 + 0.5], np.float32)
    nx = np.array([nxy, nxy, nz])
    spread = float(rng.choice([0.02, 0.1, 0.6]))                       # fraction of the plane the points fall into
    centre = (rng.rand(1, 1, 3).astype(np.float32) - 0.5) * np.array([nxy * 0.3, nxy * 0.3, nz * 0.5], np.float32)
    geom = centre + (rng.rand(B, P, 3).astype(np.float32) - 0.5) * np.array([nxy * 0.5 * spread, nxy * 0.5 * spread, nz * 1.2], np.float32)
    if P >= 128:
        geom[0, :P // 2] = geom[0, 0]                                  # half of sample 0 piled into ONE voxel
    x = rng.randn(B, P, C)
    return geom, x, dx, bx, nx, C


def _check_random_pool(seed):
    geom, x, dx, bx, nx, C = _random_pool_case(seed)
    out, xg = pool(geom, x, dx, bx, nx, requires_grad=True)
    ref, kept = so.voxel_pooling(geom, x, dx, bx, nx)
    assert hp.rel_err(out, ref) <= 1e-12, (seed, hp.rel_err(out, ref))
    gout = np.random.RandomState(seed).randn(*out.shape)
    out.backward(torch.as_tensor(gout).to(DEV))
    assert np.array_equal(xg.grad.cpu().numpy(), so.voxel_pooling_grad(geom, gout, dx, bx, nx, C)), seed


@pytest.mark.parametrize('seed', range(16))
def test_random_clustered_pooling_vs_oracle(seed):
    """float64 pooling of random, clustered point sets vs the exact sums of the oracle (forward <= 1e-12, backward bit-exact):
    piles of up to 2000 points in one voxel, channel counts around 64, planes that are no multiple of the tile, 1..3 slabs."""
    _check_random_pool(seed)


@pytest.mark.parametrize('seed', range(12))
def test_random_clustered_pooling_float32_rows_in_flight_vs_oracle(seed):
    """The float32 forward of round 4 (sixteen lanes per point, a tile's occupied voxels dealt to its rows by rank, eight rows in flight)
    on random clustered problems -- piles of up to 2000 points in one voxel, one to three chunks per row and many, ragged last tiles,
    C = 4 ... 132, empty samples -- against the oracle's exact (float64) sums: float32 accumulation of a pile in ascending point order
    stays within 1e-5 of the largest sum; a dropped or doubled point would be 1e-2."""
    rng = np.random.RandomState(7000 + seed)
    B = int(rng.randint(1, 4))
    C = int(rng.choice([4, 16, 64, 68, 132]))
    nxy, nz = 2 * int(rng.randint(3, 36)), int(rng.randint(1, 4))                      # plane % 4 == 0: the sixteen-lane kernels' shapes
    P = int(rng.choice([1, 63, 64, 65, 128, 500, 1500, 4000]))
    dx = np.array([0.5, 0.5, 1.0], np.float32)
    bx = np.array([-nxy * 0.25 + 0.25, -nxy * 0.25 + 0.25, -nz * 0.5 + 0.5], np.float32)
    nx = np.array([nxy, nxy, nz])
    spread = float(rng.choice([0.02, 0.1, 0.6]))
    centre = (rng.rand(1, 1, 3).astype(np.float32) - 0.5) * np.array([nxy * 0.3, nxy * 0.3, nz * 0.5], np.float32)
    geom = centre + (rng.rand(B, P, 3).astype(np.float32) - 0.5) * np.array([nxy * 0.5 * spread, nxy * 0.5 * spread, nz * 1.2], np.float32)
    if P >= 128:
        geom[0, :P // 2] = geom[0, 0]                                                  # half of sample 0 piled into ONE voxel
    if B > 1 and seed % 3 == 0:
        geom[-1] += 1000.0                                                             # an empty sample
    x = rng.randn(B, P, C).astype(np.float32)
    out, _ = pool(geom, x, dx, bx, nx)
    assert out.dtype == torch.float32
    ref, kept = so.voxel_pooling(geom, x.astype(np.float64), dx, bx, nx)
    assert hp.rel_err(out, ref) <= 1e-5, (seed, hp.rel_err(out, ref))
    out2, _ = pool(geom, x, dx, bx, nx)
    assert torch.equal(out, out2)


_QUAD_CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from tests.test_splat_gpu import _quad_cases
torch.save(_quad_cases(), %r)
'''


def _quad_cases():
    """float32 pooling problems the 16-lanes-per-point forward covers (C % 4 == 0, plane % 4 == 0): config-4 shapes, a ragged last
    tile (plane 2500), C = 80 (a second, partial channel chunk), nz = 3, a pile of 300 points in one voxel, an empty sample; plain
    splat and the fused lift-splat."""
    from monoforce_amd import splat
    outs = []
    m, geom, x = _c4_problem(B=2, C=64, seed=11)
    outs.append(pool(geom, x.astype(np.float32), m.dx, m.bx, m.nx)[0].cpu())
    for C, nxy, nz in ((80, 50, 1), (64, 40, 3), (8, 16, 1), (4, 6, 2)):
        rng = np.random.RandomState(C + nxy)
        B, P = 2, 900
        dx = np.array([0.5, 0.5, 1.0], np.float32)
        bx = np.array([-nxy * 0.25 + 0.25, -nxy * 0.25 + 0.25, -nz * 0.5 + 0.5], np.float32)
        nx = np.array([nxy, nxy, nz])
        geom = (rng.rand(B, P, 3).astype(np.float32) - 0.5) * np.array([nxy * 0.6, nxy * 0.6, nz * 1.2], np.float32)
        geom[0, :300] = np.array([0.1, 0.1, 0.0], np.float32)
        geom[1, :] += 1000.0
        outs.append(pool(geom, rng.randn(B, P, C).astype(np.float32), dx, bx, nx)[0].cpu())
    for shape in (dict(B=1, N=4, D=59, fH=16, fW=32, C=64, nx=256, nz=1), dict(B=2, N=3, D=7, fH=5, fW=9, C=80, nx=40, nz=2)):
        g = torch.Generator().manual_seed(5)
        B, N, D, fH, fW, C = (shape[k] for k in ('B', 'N', 'D', 'fH', 'fW', 'C'))
        nxy, nz = shape['nx'], shape['nz']
        geom = (torch.rand(B, N, D, fH, fW, 3, generator=g) - 0.5) * torch.tensor([nxy * 0.06, nxy * 0.06, nz * 1.3])
        dx = torch.tensor([0.05, 0.05, 1.0]); bx = torch.tensor([-nxy * 0.025 + 0.025, -nxy * 0.025 + 0.025, -nz * 0.5 + 0.5]); nx = torch.tensor([nxy, nxy, nz])
        plan = splat.SplatPlan(geom.to(DEV), dx, bx, nx)
        depth = torch.rand(B * N, D, fH, fW, generator=g).softmax(dim=1).to(DEV)
        ctx = torch.randn(B * N, C, fH, fW, generator=g).to(DEV)
        outs.append(splat._LiftPool.apply(depth, ctx, plan).cpu())
    return outs


def test_quad_forward_is_bit_identical_to_the_point_per_iteration_kernels():
    """Round 4's forward (sixteen lanes per point, a tile's voxels dealt round robin to the workgroup's sixteen rows) sums every voxel's
    points in ascending order with the products rounded on their own, like the kernels it replaces: the same bits -- checked against
    a child process that runs the old kernels (MF_SPLAT_QUAD=0)."""
    import os, subprocess, sys, tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    new = _quad_cases()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'old.pt')
        r = subprocess.run([sys.executable, '-c', _QUAD_CHILD % (repo, path)], capture_output=True, text=True, timeout=600, env=dict(os.environ, MF_SPLAT_QUAD='0'))
        assert r.returncode == 0, r.stderr[-2000:]
        old = torch.load(path)
    assert len(new) == len(old) == 7
    for i, (a, b) in enumerate(zip(new, old)):
        assert a.dtype == torch.float32 and float(b.abs().max()) > 0
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
