"""HBM ceilings as plain torch ops see them (GPU box): fill, copy, read-reduce at a few sizes."""
import torch, time
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for mb in (16, 64, 256, 1024):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x)); tr = t(lambda: x.sum())
    print(f"{mb:5d} MiB: fill {tf:7.1f} us = {mb*1.048576/tf*1e3:6.0f} GB/s | copy {tc:7.1f} us = {2*mb*1.048576/tc*1e3:6.0f} GB/s | sum {tr:7.1f} us = {mb*1.048576/tr*1e3:6.0f} GB/s")
