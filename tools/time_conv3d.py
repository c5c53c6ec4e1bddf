#!/usr/bin/env python3
"""Check and time the HIP 3x3x3 conv against torch / MIOpen on the GPU at the UNet's
layer shapes (cfg-B).  usage: tools/time_conv3d.py [--check-only]"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.ops import conv3d_3x3x3

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
g = torch.Generator().manual_seed(0)
shapes = [(16, 16, (16, 200, 200)), (32, 16, (16, 200, 200)), (32, 32, (8, 100, 100)), (32, 32, (4, 50, 50)),
          (16, 32, (5, 37, 71))]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, vol in shapes:
    x = torch.randn(1, cin, *vol, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).to(dev)
    up = torch.randn(1, cout, *vol, generator=g).to(dev)
    a, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ya = F.conv3d(a, wa, padding=1); ya.backward(up)
    b, wb = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yb = conv3d_3x3x3(b, wb); yb.backward(up)
    rel = lambda p, q: float((p - q).abs().max() / q.abs().max())
    print("%2d->%2d %s  rel err: out %.1e  grad_in %.1e  grad_w %.1e" % (cin, cout, vol, rel(yb, ya), rel(b.grad, a.grad),
                                                                         rel(wb.grad, wa.grad)), flush=True)
    if "--check-only" in sys.argv:
        continue
    flops = 2.0 * 27 * cin * cout * vol[0] * vol[1] * vol[2]
    def run(f, bwd):
        xx, ww = x.detach().requires_grad_(bwd), w.detach().requires_grad_(bwd)
        y = f(xx, ww)
        if bwd:
            y.backward(up)
    for name, f in (("hip", conv3d_3x3x3), ("miopen", lambda p, q: F.conv3d(p, q, padding=1))):
        tf, tfb = timeit(lambda: run(f, False)), timeit(lambda: run(f, True))
        print("        %-7s forward %7.1f us (%5.1f TFLOP/s)   forward+backward %8.1f us (%5.1f TFLOP/s)" %
              (name, tf, flops / tf / 1e6, tfb, 3 * flops / tfb / 1e6), flush=True)
