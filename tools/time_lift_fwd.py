#!/usr/bin/env python3
"""Development aid (GPU box): the lift forward alone, no-grad and grad mode, HIP events per kernel + end to end.
usage: tools/time_lift_fwd.py [cfg] [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd import _capi, synthetic
from vampire_amd.ops import HotPath
from vampire_amd.geometry import lift_matrices
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
hp = HotPath(cfg, dev)
s2e, K, ida = synthetic.camera_rig(cfg, B)
lm = lift_matrices(s2e, K, ida, synthetic.bda_matrix(B)).to(dev)
depth, feat = synthetic.lift_inputs(cfg, B, device=dev)
for grad in (False, True):
    depth.requires_grad_(grad); feat.requires_grad_(grad)
    with torch.set_grad_enabled(grad):
        for _ in range(5):
            hp.lift(depth, feat, lm)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            hp.lift(depth, feat, lm)
        e.record(); torch.cuda.synchronize()
        tot = s.elapsed_time(e) / 50 * 1e3
        _capi.profile_select(None); _capi.profile_enable(True)
        for _ in range(20):
            hp.lift(depth, feat, lm)
        torch.cuda.synchronize()
        _capi.profile_enable(False)
        ks = {k: round(ms / 20 * 1e3, 1) for k, (n, ms) in _capi.profile_read().items()}
    print(f"grad={grad}: end to end {tot:.1f} us/call; kernels {ks}")
