#!/usr/bin/env python3
"""Time the occupancy / lidar-point resampling at cfg-B on the GPU: HIP kernels vs
torch's F.grid_sample (aten) on the same device, forward and forward+backward."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import CFG_B as cfg
from vampire_amd.geometry import make_occ_coords
from vampire_amd.ops import HotPath
from vampire_amd import synthetic

dev = torch.device("cuda:0")
hp = HotPath(cfg, dev)
dens, sem, _, _ = (t.to(dev) for t in synthetic.render_inputs(cfg, 1, seed=4))
bda = synthetic.bda_matrix(1, rot_deg=5.0).to(dev)
occ = make_occ_coords().to(dev)
beta = torch.tensor(0.1, device=dev, requires_grad=True)
lo = torch.tensor([b[0] for b in (cfg.x_bound_seg, cfg.y_bound_seg, cfg.z_bound_seg)], device=dev)
span = torch.tensor([b[1] - b[0] for b in (cfg.x_bound_seg, cfg.y_bound_seg, cfg.z_bound_seg)], device=dev)


def hip(bwd):
    s = sem.detach().requires_grad_(bwd); d = dens.detach().requires_grad_(bwd)
    a, b = hp.occupancy_queries(s, d, occ, bda, beta)
    if bwd:
        (a.sum() + b.sum()).backward()


def aten(bwd):
    s = sem.detach().requires_grad_(bwd); d = dens.detach().requires_grad_(bwd)
    pts = torch.matmul(occ.reshape(1, -1, 3), bda[:, :3, :3].transpose(1, 2)).reshape(1, *occ.shape)
    g = (pts - lo) / span * 2 - 1
    a = F.grid_sample(s, g, padding_mode="border", align_corners=True)
    sig = (1 / (beta.abs() + 1e-4)) * (0.5 + 0.5 * torch.sign(d - cfg.sdf_bias) *
                                        torch.expm1(-(d - cfg.sdf_bias).abs() / (beta.abs() + 1e-4)))
    b = F.grid_sample(sig, g, align_corners=True)
    if bwd:
        (a.sum() + b.sum()).backward()


def timeit(fn, *a, n=20):
    for _ in range(5):
        fn(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(*a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


V = cfg.vZ * cfg.vY * cfg.vX
P = occ.numel() // 3
alg_fwd = 4 * (cfg.num_classes + 1) * (V + P) + 12 * P
print("occupancy queries, cfg-B, 1 sample: %d points, %d + 1 channels; algorithmic %.1f MB forward" %
      (P, cfg.num_classes, alg_fwd / 1e6))
for name, fn in (("hip", hip), ("aten", aten)):
    f, fb = timeit(fn, False), timeit(fn, True)
    print("%-5s forward %7.1f us (%.0f GB/s of algorithmic bytes)   forward+backward %7.1f us" %
          (name, f, alg_fwd / f / 1e3, fb))
