#!/usr/bin/env python3
"""Profiling target: the forward pair captured into a HIP graph (camera / BEV branch on two streams), replayed N times."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev)
batch = SyntheticBatch(cfg, 1, dev, seed=0)
model.hp.impl["fwd_overlap"] = True
with torch.no_grad():
    def fwd():
        return model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
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
        fwd()
    torch.cuda.synchronize()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
