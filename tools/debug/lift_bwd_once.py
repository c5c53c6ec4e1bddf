import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev); data = SyntheticBatch(cfg, 1, dev)
vox = model.hp.lift(data.depth, data.feat, data.lift_mats)
vox.sum().backward(); torch.cuda.synchronize()
print("done", float(data.depth.grad.abs().sum()), float(data.feat.grad.abs().sum()))
