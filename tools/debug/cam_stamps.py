#!/usr/bin/env python3
"""Debug aid (GPU box): per-tile phase stamps of the one-kernel camera forward.
Build the diagnostic library first:  tools/ablate.sh render_cam_direct.hip stamps=-DVAMP_DIRECT_STAMPS
Run:  VAMPIRE_HIP_LIB=vampire_amd/_lib/abl_stamps.so python tools/debug/cam_stamps.py [noert]"""
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
hp.impl["ert"] = not (len(sys.argv) > 1 and sys.argv[1] == "noert")
with torch.no_grad():
    for _ in range(5):
        hp.render(*batch.vols, model.beta, render_mats=batch.render_mats)
    torch.cuda.synchronize()
lib = _capi.load()
n = 1056
buf = (C.c_longlong * (n * 8))()
lib.vamp_debug_direct_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.vamp_debug_direct_stamps(buf, n * 8) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(n, 8)
t = a[:, :6].astype(np.float64)
t0 = t[:, 0].min()
ph = np.diff(t, axis=1) / 100.0            # s_memtime ticks at 100 MHz -> us
start = (t[:, 0] - t0) / 100.0
end = (t[:, 5] - t0) / 100.0
A, Seff, Se = a[:, 6] >> 32, (a[:, 6] >> 16) & 0xffff, a[:, 6] & 0xffff
names = ["plan", "density", "scan", "gather", "merge"]
print("tiles", n, "kernel span (first start -> last end) %.1f us" % end.max())
print("start offsets: median %.1f  p90 %.1f  max %.1f us" % (np.median(start), np.percentile(start, 90), start.max()))
for i, nm in enumerate(names):
    print("%-8s mean %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (nm, ph[:, i].mean(), np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max()))
tot = (t[:, 5] - t[:, 0]) / 100.0
print("tile total: mean %.2f median %.2f p90 %.2f max %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
print("A (active indices): mean %.1f max %d;  S_eff mean %.1f;  Se mean %.1f max %d" % (A.mean(), A.max(), Seff.mean(), Se.mean(), Se.max()))
order = np.argsort(-end)[:12]
print("last tiles to finish: blk start end | plan dens scan gath merge | A S_eff Se")
for b in order:
    print("%5d %6.1f %6.1f | %s | %d %d %d" % (b, start[b], end[b], " ".join("%5.1f" % x for x in ph[b]), A[b], Seff[b], Se[b]))
# per tile row (8 rows of 22 tiles per camera at cfg-B): where the long tiles sit
blk = np.arange(n)
per_xcd = (n + 7) // 8
tile = (blk & 7) * per_xcd + (blk >> 3)
ty = (tile % 176) // 22
cam = tile // 176
print("tile row: mean total us | mean start us | mean Se")
for r in range(8):
    m = ty == r
    print("  row %d: %6.2f | %6.2f | %5.1f" % (r, tot[m].mean(), start[m].mean(), Se[m].mean()))
print("camera: mean total us")
for c in range(6):
    print("  cam %d: %6.2f" % (c, tot[cam == c].mean()))
