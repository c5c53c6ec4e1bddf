#!/usr/bin/env python3
"""Debug aid: does the captured forward (camera / BEV branch on two streams) reproduce the eager one?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch
cfg = PRESETS["B"]
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev)
batch = SyntheticBatch(cfg, 1, dev, seed=0)
hp = model.hp
with torch.no_grad():
    def fwd():
        return model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
    vox0, outs0 = fwd()
    ref = [vox0.clone()] + [o.clone() for o in outs0]
    hp.impl["fwd_overlap"] = True
    vox1, outs1 = fwd()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip([vox1] + list(outs1), ref)):
        print("eager overlap", i, float((a - b).abs().max()), float(b.abs().max()))
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        vox, outs = fwd()
    torch.cuda.synchronize()
    for k in range(3):
        g.replay()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip([vox] + list(outs), ref)):
            print("replay", k, i, float((a - b).abs().max()), float(b.abs().max()), bool(torch.isfinite(a).all()))
