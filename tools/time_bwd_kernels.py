#!/usr/bin/env python3
"""Development aid (GPU box): per-kernel HIP-event times of one training step (single stream)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd import _capi
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev)
if os.environ.get("VAMP_OVERLAP", "1") == "0":
    model.hp.impl["overlap"] = False
import torch as _t
data = SyntheticBatch(cfg, B, dev, dtype=(_t.bfloat16 if os.environ.get("VAMP_DTYPE") == "bf16" else _t.float32))
for _ in range(5):
    model.zero_grad(set_to_none=True); train_step(model, data)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    model.zero_grad(set_to_none=True); train_step(model, data)
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
_capi.profile_select(None); _capi.profile_enable(True)
for _ in range(5):
    model.zero_grad(set_to_none=True); train_step(model, data)
torch.cuda.synchronize()
_capi.profile_enable(False)
tot = 0
for k, (n, ms) in sorted(_capi.profile_read().items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} {ms / 5 * 1e3:8.1f} us/step  ({n // 5} launches)")
    tot += ms / 5 * 1e3
print(f"  kernel sum {tot:.1f} us")
