#!/usr/bin/env python3
"""Time individual backward kernels via the in-library event timer (dev aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd import _capi
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev); batch = SyntheticBatch(cfg, B, dev)
for _ in range(3): train_step(model, batch)
torch.cuda.synchronize(); _capi.profile_enable(True)
for _ in range(5): train_step(model, batch)
torch.cuda.synchronize(); _capi.profile_enable(False)
print({k: round(ms / n * 1e3, 1) for k, (n, ms) in sorted(_capi.profile_read().items())})
