import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd import synthetic
from vampire_amd.ops import HotPath
from vampire_amd.geometry import lift_matrices
cfg = PRESETS["B"]; dev = torch.device("cuda:0")
hp = HotPath(cfg, dev)
s2e, K, ida = synthetic.camera_rig(cfg, 1)
lm = lift_matrices(s2e, K, ida, synthetic.bda_matrix(1)).to(dev)
depth, feat = synthetic.lift_inputs(cfg, 1, device=dev)
with torch.no_grad():
    for _ in range(30):
        hp.lift(depth, feat, lm)
        if os.environ.get("VAMPIRE_HIP_LIB") is None:
            hp.lift_cull_words(lm)
torch.cuda.synchronize()
