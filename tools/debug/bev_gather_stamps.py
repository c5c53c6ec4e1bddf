#!/usr/bin/env python3
"""Debug aid (GPU box): per-workgroup phase stamps of the composited BEV column gather (wall_clock64, 100 MHz).
Build the diagnostic library first:  tools/ablate.sh render_bev.hip stamps=-DVAMP_COMP_STAMPS
Run:  VAMPIRE_HIP_LIB=vampire_amd/_lib/abl_stamps.so python tools/debug/bev_gather_stamps.py"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch
from vampire_amd import _capi
cfg = PRESETS["B"]
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev)
batch = SyntheticBatch(cfg, 1, dev, seed=0)
hp = model.hp
hp.impl["overlap"] = False
vols = [v.clone().requires_grad_(True) for v in batch.vols]
for _ in range(4):
    outs = hp.render(*vols, model.beta, render_mats=batch.render_mats)
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
torch.cuda.synchronize()
lib = _capi.load()
n = 4096
buf = (C.c_longlong * (n * 8))()
lib.vamp_debug_comp_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.vamp_debug_comp_stamps(buf, n * 8) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(n, 8).astype(np.float64)
a = a[a[:, 0] > 0]
full = a[a[:, 4] > 0]
t0 = a[:, 0].min()
print("workgroups stamped", len(a), "with all phases", len(full))
end = np.where(a[:, 4] > 0, a[:, 4], a[:, 0])
print("span (first start -> last end) %.1f us" % ((end.max() - t0) / 100.0))
start = (a[:, 0] - t0) / 100.0
print("start offsets: p10 %.1f median %.1f p90 %.1f max %.1f us" % tuple(np.percentile(start, [10, 50, 90, 100])))
names = ["tables", "loads issued + factors", "heights", "last stores"]
ph = np.diff(full[:, :5], axis=1) / 100.0
for i, x in enumerate(names):
    print("  %-24s mean %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (x, ph[:, i].mean(), np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max()))
tot = (full[:, 4] - full[:, 0]) / 100.0
print("  total: mean %.2f median %.2f p90 %.2f max %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
