"""What is left of the UNet under bf16 autocast: torch profiler, top kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.backbone import Unet3D
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
net = Unet3D(16, 16).to(dev)
x = torch.randn(1, 16, 16, 200, 200, device=dev, requires_grad=True)
def step():
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    y.float().mean().backward()
for _ in range(4):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
allr = prof.key_averages()
print("total device time %.1f us" % sum(e.self_device_time_total for e in allr))
rows = sorted(allr, key=lambda e: -e.self_device_time_total)[:14]
for e in rows:
    print("%-90s %4d %9.1f us" % (e.key[:90], e.count, e.self_device_time_total))
