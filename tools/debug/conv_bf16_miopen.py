"""How MIOpen does on the UNet's 3x3x3 layers in bf16 (what autocast hands it), beside fp32 MIOpen and the HIP fp32 kernel."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.ops import conv3d_3x3x3, conv3d_bf16
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
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
for cin, cout, vol in [(16, 16, (16, 200, 200)), (16, 32, (16, 200, 200)), (32, 16, (16, 200, 200)), (32, 32, (10, 128, 128))]:
    x = torch.randn(1, cin, *vol, device=dev); w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    up = torch.randn(1, cout, *vol, device=dev)
    for name, dt, f in (("miopen bf16", torch.bfloat16, lambda p, q: F.conv3d(p, q, padding=1)),
                        ("miopen bf16 channels_last_3d", "cl", lambda p, q: F.conv3d(p, q, padding=1)),
                        ("miopen fp32", torch.float32, lambda p, q: F.conv3d(p, q, padding=1)),
                        ("hip fp32", torch.float32, conv3d_3x3x3), ("hip bf16", torch.bfloat16, conv3d_bf16)):
        if name.startswith("hip") and cin not in (16, 32):
            continue
        if dt == "cl":
            xx0, ww0, uu = x.bfloat16().contiguous(memory_format=torch.channels_last_3d), w.bfloat16().contiguous(memory_format=torch.channels_last_3d), up.bfloat16().contiguous(memory_format=torch.channels_last_3d)
        else:
            xx0, ww0, uu = x.to(dt), w.to(dt), up.to(dt)
        def run(bwd):
            xx, ww = xx0.detach().requires_grad_(bwd), ww0.detach().requires_grad_(bwd)
            y = f(xx, ww)
            if bwd:
                y.backward(uu)
        tf, tfb = timeit(lambda: run(False)), timeit(lambda: run(True))
        print("%2d->%2d %-16s %-30s forward %8.1f us   forward+backward %9.1f us" % (cin, cout, vol, name, tf, tfb), flush=True)
