"""CPU oracle for the LSS BEV voxel pooling  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Restates `LiftSplatShoot.voxel_pooling` (`/root/reference/monoforce/src/monoforce/models/terrain_encoder/lss.py:238-280`):
voxel index = trunc((geom - (bx - dx/2)) / dx) in FLOAT32 (bit-identical indices to the reference), in-bounds mask,
then the per-voxel sum -- accumulated in float64 ("exact" sums) instead of the reference's float32 prefix-sum trick
(`terrain_encoder/utils.py:144-181`), whose ~1e-3 relative error makes it unusable as a parity target.
Pinned by tests/golden/lss.npz: indices / kept mask against the reference's, sums against both the reference's float32
output (to its own accuracy) and the float64 exact sums computed next to it.
"""
import numpy as np


def voxel_index(geom, dx, bx):
    """geom [..., 3] float32 -> int64 voxel indices [..., 3] with the reference's float32 arithmetic (lss.py:246)."""
    geom = np.asarray(geom, np.float32)
    dx, bx = np.asarray(dx, np.float32), np.asarray(bx, np.float32)
    off = (bx - dx / np.float32(2.0)).astype(np.float32)
    v = ((geom - off).astype(np.float32) / dx).astype(np.float32)
    return np.trunc(v).astype(np.int64)


def voxel_pooling(geom, x, dx, bx, nx, acc_dtype=np.float64):
    """geom [B,...,3], x [B,...,C] -> [B, nz*C, nx, ny] per-voxel sums (channel index iz*C + c, lss.py:274-278)."""
    B, Cc = x.shape[0], x.shape[-1]
    nX, nY, nZ = (int(v) for v in nx)
    idx = voxel_index(geom, dx, bx).reshape(B, -1, 3)
    xf = np.asarray(x).reshape(B, -1, Cc).astype(acc_dtype)
    kept = ((idx[..., 0] >= 0) & (idx[..., 0] < nX) & (idx[..., 1] >= 0) & (idx[..., 1] < nY)
            & (idx[..., 2] >= 0) & (idx[..., 2] < nZ))
    out = np.zeros((B, nZ, nX * nY, Cc), acc_dtype)
    for b in range(B):
        k = kept[b]
        lin = idx[b, k, 0] * nY + idx[b, k, 1]
        np.add.at(out[b], (idx[b, k, 2], lin), xf[b, k])
    out = out.reshape(B, nZ, nX, nY, Cc).transpose(0, 1, 4, 2, 3).reshape(B, nZ * Cc, nX, nY)
    return out, kept


def voxel_pooling_grad(geom, gout, dx, bx, nx, C):
    """QuickCumsum.backward (utils.py:174-181): grad_x[p] = gout[voxel(p)] for kept points, 0 otherwise."""
    gout = np.asarray(gout)
    B = gout.shape[0]
    nX, nY, nZ = (int(v) for v in nx)
    idx = voxel_index(geom, dx, bx).reshape(B, -1, 3)
    kept = ((idx[..., 0] >= 0) & (idx[..., 0] < nX) & (idx[..., 1] >= 0) & (idx[..., 1] < nY)
            & (idx[..., 2] >= 0) & (idx[..., 2] < nZ))
    g = gout.reshape(B, nZ, C, nX, nY)
    gx = np.zeros((B, idx.shape[1], C), gout.dtype)
    for b in range(B):
        k = kept[b]
        gx[b, k] = g[b, idx[b, k, 2], :, idx[b, k, 0], idx[b, k, 1]]
    return gx
