#!/usr/bin/env python3
"""Dev aid (GPU box; library built with -DVAMP_RAY_STAMPS: tools/ablate.sh render_bwd_ray.hip stamps=-DVAMP_RAY_STAMPS):
phase timeline of the camera backward's per-ray pass (100 MHz wall clock)."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
cfg = PRESETS["B"]; dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev); model.hp.impl["overlap"] = False
data = SyntheticBatch(cfg, 1, dev)
for _ in range(3):
    model.zero_grad(set_to_none=True); train_step(model, data)
torch.cuda.synchronize()
n = 1056
buf = np.zeros((n, 8), dtype=np.int64)
lib = model.hp.lib
lib.vamp_debug_ray_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.vamp_debug_ray_stamps(buf.ctypes.data, n) == 0
t0 = buf[:, 0].min(); span = (buf[:, 5].max() - t0) / 100.0
print("kernel span %.1f us; per WG total mean %.1f us p90 %.1f" % (span, ((buf[:, 5] - buf[:, 0]) / 100.0).mean(), np.percentile((buf[:, 5] - buf[:, 0]) / 100.0, 90)))
names = ["setup + G loads", "loop 1 (samples, records)", "merge", "loop 2 (weights)", "beta reduce"]
for k, nm in enumerate(names):
    d = (buf[:, k + 1] - buf[:, k]) / 100.0
    print("  %-28s mean %6.2f us  p90 %6.2f" % (nm, d.mean(), np.percentile(d, 90)))
st = (buf[:, 0] - t0) / 100.0
print("starts by decile:", np.histogram(st, bins=10, range=(0, span))[0].tolist())
tot = (buf[:, 5] - buf[:, 0]) / 100.0
print("per-WG total: max %.1f us; WGs > 20 us: %d; Se mean %.1f max %d; corr(Se, total) %.2f" % (tot.max(), int((tot > 20).sum()), buf[:, 6].mean(), buf[:, 6].max(), np.corrcoef(buf[:, 6], tot)[0, 1]))
order = np.argsort(-tot)[:8]
print("slowest WGs (idx, total us, Se, start us):", [(int(i), round(float(tot[i]), 1), int(buf[i, 6]), round(float(st[i]), 1)) for i in order])
ends = (buf[:, 5] - t0) / 100.0
print("ends by decile:", np.histogram(ends, bins=10, range=(0, span))[0].tolist())
