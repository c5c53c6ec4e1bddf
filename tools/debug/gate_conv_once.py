"""The fused gate + 1x1 conv, forward + backward, a few times (for rocprofv3 / ablations)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import CFG_B as cfg
from vampire_amd.ops import HotPath
dev = torch.device("cuda:0")
hp = HotPath(cfg, dev)
side = int(sys.argv[1]) if len(sys.argv) > 1 else cfg.oY
g = torch.Generator().manual_seed(0)
vo = torch.randn(1, 16, 10, side, side, generator=g).to(dev).requires_grad_(True)
vd = torch.rand(1, 1, 10, side, side, generator=g).to(dev).requires_grad_(True)
w = (torch.randn(80, 160, 1, 1, generator=g) * 0.1).to(dev).requires_grad_(True)
b = torch.randn(80, generator=g).to(dev).requires_grad_(True)
go = torch.randn(1, 80, side, side, generator=g).to(dev)
for _ in range(20):
    hp.gate_conv1x1(vo, vd, w, b).backward(go)
torch.cuda.synchronize()
