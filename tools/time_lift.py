#!/usr/bin/env python3
"""Development aid (GPU box): per-kernel HIP-event times of the lift forward + backward alone.
usage: tools/time_lift.py [cfg] [batch] [logits]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd import _capi, synthetic
from vampire_amd.ops import HotPath
from vampire_amd.geometry import lift_matrices
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
logits = len(sys.argv) > 3 and sys.argv[3] == "logits"
dev = torch.device("cuda:0")
hp = HotPath(cfg, dev)
s2e, K, ida = synthetic.camera_rig(cfg, B)
lm = lift_matrices(s2e, K, ida, synthetic.bda_matrix(B)).to(dev)
depth, feat = synthetic.lift_inputs(cfg, B, device=dev)
if logits:
    depth = torch.randn_like(depth) * 3
depth.requires_grad_(True); feat.requires_grad_(True)
go = torch.randn(B, cfg.mid_channels, cfg.vZ, cfg.vY, cfg.vX, device=dev)


def step():
    depth.grad = None; feat.grad = None
    out = hp.lift_logits(depth, feat, lm) if logits else hp.lift(depth, feat, lm)
    out.backward(go)


for _ in range(5):
    step()
torch.cuda.synchronize()
_capi.profile_select(None); _capi.profile_enable(True)
for _ in range(10):
    step()
torch.cuda.synchronize()
_capi.profile_enable(False)
tot = 0
for k, (n, ms) in sorted(_capi.profile_read().items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} {ms / 10 * 1e3:8.1f} us/step  ({n // 10} launches)")
    tot += ms / 10 * 1e3
print(f"  kernel sum {tot:.1f} us")
