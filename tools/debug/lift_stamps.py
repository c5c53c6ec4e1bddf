#!/usr/bin/env python3
"""Dev aid (GPU box, library built with -DVAMP_LIFT_STAMPS (tools/ablate.sh lift_bwd_cell.hip stamps=-DVAMP_LIFT_STAMPS)): phase timeline of the lift backward strip gather."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd import _capi, synthetic
from vampire_amd.ops import HotPath
from vampire_amd.geometry import lift_matrices
cfg = PRESETS["B"]; dev = torch.device("cuda:0"); hp = HotPath(cfg, dev)
s2e, K, ida = synthetic.camera_rig(cfg, 1)
lm = lift_matrices(s2e, K, ida, synthetic.bda_matrix(1)).to(dev)
depth, feat = synthetic.lift_inputs(cfg, 1, device=dev)
depth.requires_grad_(True); feat.requires_grad_(True)
go = torch.randn(1, cfg.mid_channels, cfg.vZ, cfg.vY, cfg.vX, device=dev)
for _ in range(3):
    depth.grad = None; feat.grad = None
    hp.lift(depth, feat, lm).backward(go)
torch.cuda.synchronize()
n = 6 * cfg.fH * ((cfg.fW + 15) // 16)
buf = np.zeros((n, 8), dtype=np.int64)
lib = hp.lib
lib.vamp_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.vamp_debug_read_stamps(buf.ctypes.data, n) == 0
r0 = buf[:, 6].min(); span = buf[:, 7].max() - r0
print("WGs", n, "kernel span %.1f us (100 MHz clock)" % (span / 100.0))
dur = buf[:, 5] - buf[:, 0]; rdur = (buf[:, 7] - buf[:, 6]) / 100.0
print("per WG us: mean %.2f p50 %.2f p90 %.2f max %.2f; ticks per us %.0f" % (rdur.mean(), *np.percentile(rdur, [50, 90]), rdur.max(), dur.sum() / rdur.sum()))
print("mean concurrency %.0f WGs" % (rdur.sum() / (span / 100.0)))
ph0 = buf[:, 1] - buf[:, 0]; stage = buf[:, 2]; cons = buf[:, 3]; epi = buf[:, 5] - buf[:, 4]
other = dur - ph0 - stage - cons - epi
for name, v in (("phase0", ph0), ("stage", stage), ("consume", cons), ("epilogue", epi), ("other", other)):
    print("  %-9s mean %8.0f  p50 %8.0f p90 %8.0f max %8.0f  share %.2f" % (name, v.mean(), *np.percentile(v, [50, 90]), v.max(), v.sum() / dur.sum()))
st = (buf[:, 6] - r0)
print("starts by decile of span:", np.histogram(st, bins=10, range=(0, span))[0].tolist())
print("ends by decile of span:", np.histogram(buf[:, 7] - r0, bins=10, range=(0, span))[0].tolist())
# concurrency over time
ts = np.linspace(0, span, 21)
print("running at t:", [int(((st <= t) & ((buf[:, 7] - r0) > t)).sum()) for t in ts])
