import os, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "inproc":
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")
print("env at start:", os.environ.get("MIOPEN_FIND_MODE"))
import torch
torch.backends.cudnn.benchmark = True
x = torch.randn(6, 64, 64, 176, device="cuda"); w = torch.randn(64, 64, 3, 3, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
y = torch.nn.functional.conv2d(x, w, padding=1); torch.cuda.synchronize()
print("first conv2d: %.2f s" % (time.perf_counter() - t0))
x3 = torch.randn(1, 16, 16, 200, 200, device="cuda"); w3 = torch.randn(16, 16, 3, 3, 3, device="cuda")
t0 = time.perf_counter(); y = torch.nn.functional.conv3d(x3, w3, padding=1); torch.cuda.synchronize()
print("first conv3d: %.2f s" % (time.perf_counter() - t0))
