#!/usr/bin/env python3
"""Time the BEVDepth-style voxel pooling at the upstream shape: 6 cameras x 112 depth
planes x 16 x 44 frustum points x 80 channels into a 128 x 128 BEV grid; HBM roofline on the algorithmic
bytes (features read once, output written once), against an aten index_add_ of the same definition."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.ops import voxel_pooling
dev = torch.device("cuda:0")
B, N, D, H, W, C, vn = 1, 6, 112, 16, 44, 80, (128, 128, 1)
g = torch.Generator().manual_seed(0)
geom = torch.stack([torch.randint(-8, vn[0] + 8, (B, N, D, H, W), generator=g), torch.randint(-8, vn[1] + 8, (B, N, D, H, W), generator=g),
                    torch.zeros(B, N, D, H, W, dtype=torch.long)], -1).to(dev)
feat = torch.randn(B, N, D, H, W, C, generator=g).to(dev)
go = torch.randn(B, C, vn[1], vn[0], generator=g).to(dev)


def hip(bwd):
    f = feat.detach().requires_grad_(bwd)
    y = voxel_pooling(geom, f, vn)
    if bwd:
        y.backward(go)


def aten(bwd):
    f = feat.detach().requires_grad_(bwd)
    gx, gy, gz = geom[..., 0].reshape(-1), geom[..., 1].reshape(-1), geom[..., 2].reshape(-1)
    ok = (gx >= 0) & (gx < vn[0]) & (gy >= 0) & (gy < vn[1]) & (gz >= 0) & (gz < vn[2])
    cell = (gy * vn[0] + gx).clamp(0, vn[0] * vn[1] - 1)
    y = torch.zeros(vn[0] * vn[1], C, device=dev).index_add_(0, cell, f.reshape(-1, C) * ok[:, None])
    if bwd:
        y.backward(go.permute(0, 2, 3, 1).reshape(-1, C))


def timeit(fn, *a, n=30):
    for _ in range(5):
        fn(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(*a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


nb = feat.numel() * 4 + B * vn[0] * vn[1] * C * 4 + geom.numel() * 4
print("voxel pooling, %d points x %d channels -> %d x %d cells (%.0f MB algorithmic)" % (feat.numel() // C, C, vn[1], vn[0], nb / 1e6))
for name, fn in (("hip", hip), ("aten index_add_", aten)):
    f, fb = timeit(fn, False), timeit(fn, True)
    print("%-16s forward %8.1f us (%.2f TB/s, %.1f %% of 8 TB/s)   forward+backward %8.1f us" % (name, f, nb / f / 1e6, nb / f / 1e6 / 8 * 100, fb))
