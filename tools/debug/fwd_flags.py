#!/usr/bin/env python3
"""GPU box: what the training forward's extras cost in the merged render launch (HIP-event time of the launch, eager,
one stream): no-grad / training (ranks + sample rows + BEV planes) / training without the sample rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd import _capi
from vampire_amd.step import LiftRenderStep, SyntheticBatch
cfg = PRESETS["B"]; dev = torch.device("cuda:0")

def run(train, **impl):
    model = LiftRenderStep(cfg, dev)
    model.hp.impl["overlap"] = False
    model.hp.impl.update(impl)
    batch = SyntheticBatch(cfg, 1, dev)
    def fwd():
        if train:
            return model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
        with torch.no_grad():
            return model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
    for _ in range(5): fwd()
    torch.cuda.synchronize(); _capi.profile_enable(True)
    for _ in range(20): fwd()
    torch.cuda.synchronize(); _capi.profile_enable(False)
    return {k: ms / 20 * 1e3 for k, (n, ms) in _capi.profile_read().items()}

for name, r in (("no-grad", run(False)), ("training", run(True)), ("training, save_rows off", run(True, save_rows=False))):
    print("%-28s" % name, "  ".join("%s %.1f" % (k, v) for k, v in sorted(r.items()) if "render" in k or k == "aux"))
