#!/usr/bin/env python3
"""Debug aid (GPU box): per-workgroup phase stamps of the one-kernel BEV forward (wall_clock64, 100 MHz).
Build the diagnostic library first:  tools/ablate.sh render_bev_fused.hip stamps=-DVAMP_BEVF_STAMPS
Run:  VAMPIRE_HIP_LIB=vampire_amd/_lib/abl_stamps.so python tools/debug/bev_stamps.py [train]"""
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
train = len(sys.argv) > 1 and sys.argv[1] == "train"
if train:
    vols = [v.clone().requires_grad_(True) for v in batch.vols]
    for _ in range(5):                       # forward + backward, as in a step: the forward finds the caches cold
        outs = hp.render(*vols, model.beta, render_mats=batch.render_mats)
        torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
else:
    with torch.no_grad():
        for _ in range(5):
            hp.render(*batch.vols, model.beta, render_mats=batch.render_mats)
torch.cuda.synchronize()
lib = _capi.load()
n = 632
buf = (C.c_longlong * (n * 16))()
lib.vamp_debug_bevf_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.vamp_debug_bevf_stamps(buf, n * 16) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(n, 2, 8).astype(np.float64)
a = a[a[:, 0, 0] > 0]
t0 = a[:, :, 0].min()
names = ["tables", "density planes", "weights", "channels"]
print("mode", "train" if train else "forward only", "workgroups", len(a))
for w, nm in ((0, "group 0 (density + composited channels), wave 0"), (1, "group 1 (pass-through channels), wave 0")):
    t = a[:, w, :5]
    ph = np.diff(t, axis=1) / 100.0
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 4] - t0) / 100.0
    print(f"{nm}: span (first start -> last end) {end.max():.1f} us; start offsets median {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f}")
    for i, x in enumerate(names):
        print("  %-15s mean %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (x, ph[:, i].mean(), np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max()))
    tot = (t[:, 4] - t[:, 0]) / 100.0
    print("  total: mean %.2f median %.2f p90 %.2f max %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
