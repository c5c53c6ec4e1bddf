#!/usr/bin/env python3
"""Profiling target: N forward passes (lift + render) at a config; nothing else on the GPU."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd.geometry import lift_matrices, render_matrices
from vampire_amd import synthetic
from vampire_amd.ops import HotPath
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dtype = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
dev = torch.device("cuda:0")
hp = HotPath(cfg, dev)
s2e, K, ida = synthetic.camera_rig(cfg, B)
bda = synthetic.bda_matrix(B)
lm, rm = lift_matrices(s2e, K, ida, bda).to(dev), render_matrices(s2e, K, ida, bda).to(dev)
depth, feat = synthetic.lift_inputs(cfg, B, device=dev, dtype=dtype)
vols = synthetic.render_inputs(cfg, B, device=dev, dtype=dtype)
beta = torch.tensor(0.1, device=dev)
with torch.no_grad():
    for _ in range(n):
        hp.lift(depth, feat, lm)
        hp.render(*vols, beta, render_mats=rm)
torch.cuda.synchronize()
