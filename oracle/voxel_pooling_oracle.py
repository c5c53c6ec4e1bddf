"""TEST INFRASTRUCTURE ONLY -- CPU statement of the BEVDepth voxel-pooling operator that north_star names
(SURVEY 8 row a11).  **Parity unpinned**: the operator is not part of /root/reference at the pinned commit
(`setup.py` builds no extension, no file mentions it), so there is neither a golden vector nor a reference
run to pin this to; it restates the published definition (BEVDepth `voxel_pooling_forward`: for every
frustum point whose integer voxel index lies inside the grid, add its feature row to the BEV cell).
Only tests may import this module."""
import numpy as np


def voxel_pooling(geom_xyz, feats, voxel_num):
    """geom_xyz [B, P, 3] int (x, y, z), feats [B, P, C] float, voxel_num (nx, ny, nz) -> [B, C, ny, nx] float64 sums."""
    nx, ny, nz = voxel_num
    B, P, C = feats.shape
    out = np.zeros((B, ny, nx, C), dtype=np.float64)
    g = geom_xyz.astype(np.int64)
    ok = (g[..., 0] >= 0) & (g[..., 0] < nx) & (g[..., 1] >= 0) & (g[..., 1] < ny) & (g[..., 2] >= 0) & (g[..., 2] < nz)
    for b in range(B):
        m = ok[b]
        np.add.at(out[b], (g[b, m, 1], g[b, m, 0]), feats[b, m].astype(np.float64))
    return out.transpose(0, 3, 1, 2)


def voxel_pooling_backward(geom_xyz, grad_out, voxel_num):
    """grad_out [B, C, ny, nx] -> grad_feats [B, P, C]: the row of the point's cell, zeros outside the grid."""
    nx, ny, nz = voxel_num
    g = geom_xyz.astype(np.int64)
    ok = (g[..., 0] >= 0) & (g[..., 0] < nx) & (g[..., 1] >= 0) & (g[..., 1] < ny) & (g[..., 2] >= 0) & (g[..., 2] < nz)
    go = grad_out.transpose(0, 2, 3, 1)
    B = g.shape[0]
    gx, gy = np.clip(g[..., 0], 0, nx - 1), np.clip(g[..., 1], 0, ny - 1)
    rows = go[np.arange(B)[:, None], gy, gx]
    return rows * ok[..., None]
