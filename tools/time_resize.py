#!/usr/bin/env python3
"""Time the UNet's trilinear resizes (resize piece) at cfg-B on the GPU: HIP kernels vs
aten's F.interpolate on the same device (kernel times: run under tools/kstats_cmd.sh)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.ops import upsample_trilinear

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cases = [((1, 32, 8, 100, 100), (16, 200, 200)), ((1, 32, 4, 50, 50), (8, 100, 100))]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for shape, size in cases:
    x = torch.randn(shape, generator=g).to(dev)
    up = torch.randn(shape[:2] + size, generator=g).to(dev)

    def run(f, bwd):
        a = x.detach().requires_grad_(bwd)
        y = f(a)
        if bwd:
            y.backward(up)

    hip = lambda a: upsample_trilinear(a, size)
    aten = lambda a: F.interpolate(a, size, mode="trilinear", align_corners=True)
    nb_in, nb_out = x.numel() * 4, up.numel() * 4
    for name, f in (("hip", hip), ("aten", aten)):
        tf, tfb = timeit(lambda: run(f, False)), timeit(lambda: run(f, True))
        print("%s -> %s  %-5s forward %7.1f us (%.2f TB/s of %.0f MB)   forward+backward %8.1f us" %
              (tuple(shape[1:]), size, name, tf, (nb_in + nb_out) / tf / 1e6, (nb_in + nb_out) / 1e6, tfb))
