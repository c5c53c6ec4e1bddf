#!/usr/bin/env python3
"""Debug aid: is the one-kernel camera forward bit-reproducible from call to call?"""
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
if len(sys.argv) > 1:
    hp.impl["ert"] = sys.argv[1] != "noert"
with torch.no_grad():
    prev = None
    for k in range(8):
        outs = hp.render(*batch.vols, model.beta, render_mats=batch.render_mats)
        torch.cuda.synchronize()
        cur = [o.clone() for o in outs[:3]]
        if prev is not None:
            d = [(a - b).abs() for a, b in zip(cur, prev)]
            print(k, [float(x.max()) for x in d], [int((x > 0).sum()) for x in d])
            if k == 1:
                bad = (d[2] > 0).nonzero()
                print("first differing depth pixels:", bad[:10].tolist())
        prev = cur
