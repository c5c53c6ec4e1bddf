"""bf16 HIP conv forward + backward a few times (rocprofv3 / ablations): args cin cout Z Y X"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.ops import conv3d_bf16
dev = torch.device("cuda:0")
cin, cout, Z, Y, X = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 16, 16, 200, 200)))
x = torch.randn(1, cin, Z, Y, X, device=dev).bfloat16().requires_grad_(True)
w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).bfloat16().requires_grad_(True)
go = torch.randn(1, cout, Z, Y, X, device=dev).bfloat16()
for _ in range(20):
    conv3d_bf16(x, w).backward(go)
torch.cuda.synchronize()
