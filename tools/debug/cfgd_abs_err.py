#!/usr/bin/env python3
"""GPU box: worst absolute element error of the rendered tensors at cfg-D (bf16-rounded inputs) against the reference's
samples (tests/golden/cfgd_samples.npz), for the one-kernel camera forward and for copy + planned march."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vampire_amd.config import CFG_D as cfg
from vampire_amd.ops import HotPath
from vampire_amd import synthetic
NAMES = ["rgb_preds", "seg_logits_preds", "depth_preds", "bev_rgb_preds", "bev_seg_logits_preds", "bev_height_preds", "voxel_density", "voxel_output"]
dev = torch.device("cuda:0")
with np.load(os.path.join(ROOT, "tests/golden/cfgd_samples.npz")) as z:
    es = {k: torch.from_numpy(z[k]) for k in z.files}
with open(os.path.join(ROOT, "tests/golden/cfgd_checksums.json")) as f:
    ref = json.load(f)["D"]
rm = torch.tensor(ref["render_mats"], dtype=torch.float32, device=dev)
vols = [t.to(dev).bfloat16() for t in synthetic.render_inputs(cfg, 1, seed=0)]
beta = torch.tensor(0.1, device=dev)
for direct in (True, False):
    hp = HotPath(cfg, dev)
    hp.impl["cam_direct"] = direct
    with torch.no_grad():
        outs = hp.render(*vols, beta, render_mats=rm)
    res = {}
    for nm, o in zip(NAMES[:3], outs[:3]):
        want = es[nm].float(); stride = int(es[nm + "_stride"])
        got = o.float().flatten()[::stride][:want.numel()].cpu()
        res[nm] = float((got - want).abs().max())
    print("direct" if direct else "planned", {k: f"{v:.2e}" for k, v in res.items()})
