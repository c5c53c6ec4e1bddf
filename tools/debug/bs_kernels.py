#!/usr/bin/env python3
"""GPU box: per-kernel HIP-event times (one stream, eager) of the training step at batch 1 and batch 8, per sample --
with the batch's own jittered samples and with sample 0 replicated (separates what the data does from what the batch does)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd import _capi
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
cfg = PRESETS["B"]; dev = torch.device("cuda:0")

def run(B, replicate):
    model = LiftRenderStep(cfg, dev)
    model.hp.impl["overlap"] = False
    batch = SyntheticBatch(cfg, B, dev)
    if replicate:
        one = SyntheticBatch(cfg, 1, dev)
        rep = lambda t: t.detach().expand(B, *t.shape[1:]).contiguous(memory_format=torch.preserve_format) if t.shape[0] == 1 else t
        batch.depth = one.depth.detach().repeat(B, 1, 1, 1, 1).requires_grad_(True)
        f = one.feat.detach().permute(0, 1, 3, 4, 2).repeat(B, 1, 1, 1, 1).permute(0, 1, 4, 2, 3)
        batch.feat = f.requires_grad_(True)
        batch.vols = [v.detach().repeat(B, 1, 1, 1, 1).requires_grad_(True) for v in one.vols]
        batch.lift_mats = one.lift_mats.repeat(B, 1, 1, 1, 1)
        batch.render_mats = one.render_mats.repeat(B, 1, 1, 1, 1)
    for _ in range(3):
        model.zero_grad(set_to_none=True); train_step(model, batch)
    torch.cuda.synchronize(); _capi.profile_enable(True)
    for _ in range(5):
        model.zero_grad(set_to_none=True); train_step(model, batch)
    torch.cuda.synchronize(); _capi.profile_enable(False)
    return {k: ms / 5 * 1e3 / B for k, (n, ms) in _capi.profile_read().items()}

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"      # (quick: no replicated batch)
r1 = run(1, False); r8 = run(8, False); r8r = {} if quick else run(8, True)
print("%-28s %10s %10s %12s" % ("kernel (us per sample)", "batch 1", "batch 8", "8 x sample 0"))
for k in sorted(r1, key=lambda k: -r1[k]):
    print("%-28s %10.1f %10.1f %12.1f" % (k, r1[k], r8.get(k, 0), r8r.get(k, 0)))
print("%-28s %10.1f %10.1f %12.1f" % ("sum", sum(r1.values()), sum(r8.values()), sum(r8r.values())))
