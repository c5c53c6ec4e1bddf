#!/usr/bin/env python3
"""Time the Unet3D inpaintor + the three 3x3x3 head convs at cfg-B with MIOpen
through torch, in several modes.  usage: tools/time_unet3d.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.backbone import Unet3D

dev = torch.device("cuda:0")
x0 = torch.randn(1, 16, 16, 200, 200, device=dev)


def bench(tag, benchmark, dtype=None, cl=False):
    torch.backends.cudnn.benchmark = benchmark
    torch.manual_seed(0)
    net = Unet3D(16, 16).to(dev)
    x = x0.clone()
    if cl:
        net = net.to(memory_format=torch.channels_last_3d)
        x = x.contiguous(memory_format=torch.channels_last_3d)
    x.requires_grad_(True)

    def step():
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
            y = net(x)
        e1.record()
        y.float().mean().backward()

    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    f = b = 0.0
    n = 5
    for _ in range(n):
        e0.record(); step(); e2.record(); torch.cuda.synchronize()
        f += e0.elapsed_time(e1); b += e1.elapsed_time(e2)
    print("%-28s forward %8.2f ms   backward %8.2f ms" % (tag, f / n, b / n), flush=True)


# torch caches the solver picked for a shape for the life of the process: the immediate-mode run
# (367 ms backward: a 53 ms weight-gradient kernel per layer) goes last
print("VAMP_CONV3D=%s (HIP fp32 matrix-core convs for the two finer levels unless 0)" % os.environ.get("VAMP_CONV3D", "1"))
bench("fp32 benchmark", True)
bench("fp32 benchmark NDHWC", True, cl=True)
bench("bf16 autocast benchmark", True, torch.bfloat16)
bench("fp16 autocast benchmark", True, torch.float16)
if "--immediate" in sys.argv:
    bench("fp32 immediate mode", False)
