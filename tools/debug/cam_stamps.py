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
hp.impl["fwd_merged"] = False          # (the stamps live in the stand-alone kernel's translation unit)
hp.impl["cam_direct"] = True
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
# s_memtime counts shader-clock cycles (2.4 GHz here), and every XCD has its own counter: tile durations are
# comparable everywhere, start / end offsets only inside an XCD (workgroup b runs on XCD b % 8)
TICKS_PER_US = float(os.environ.get("VAMP_STAMP_TICKS_PER_US", "2400"))
xcd = np.arange(n) & 7
t0 = np.array([t[xcd == x, 0].min() for x in range(8)])[xcd]
ph = np.diff(t, axis=1) / TICKS_PER_US
start = (t[:, 0] - t0) / TICKS_PER_US
end = (t[:, 5] - t0) / TICKS_PER_US
A, Seff, Se = a[:, 6] >> 32, (a[:, 6] >> 16) & 0xffff, a[:, 6] & 0xffff
names = ["plan", "density", "scan", "gather", "merge"]
print("tiles", n, "kernel span (first start -> last end) %.1f us" % end.max())
print("start offsets: median %.1f  p90 %.1f  max %.1f us" % (np.median(start), np.percentile(start, 90), start.max()))
for i, nm in enumerate(names):
    print("%-8s mean %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (nm, ph[:, i].mean(), np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max()))
tot = (t[:, 5] - t[:, 0]) / TICKS_PER_US
print("tile total: mean %.2f median %.2f p90 %.2f max %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
print("end offsets (per XCD): median %.1f  p90 %.1f  p99 %.1f  max %.1f us" % (np.median(end), np.percentile(end, 90), np.percentile(end, 99), end.max()))
print("tiles that START after 5 us: %d (mean start %.1f us)" % ((start > 5).sum(), start[start > 5].mean() if (start > 5).any() else 0))
for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 64), (64, 200)):
    m = (Se >= lo) & (Se < hi)
    if m.any():
        print("  Se in [%3d, %3d): %4d tiles, total mean %.1f us (plan %.1f dens %.1f scan %.1f gather %.1f merge %.1f), S_eff mean %.0f" % (
            lo, hi, m.sum(), tot[m].mean(), *[ph[m, i].mean() for i in range(5)], Seff[m].mean()))
wall = a[:, 7].astype(np.uint64)
w0 = (wall >> np.uint64(32)).astype(np.float64) / 100.0
w1 = (wall & np.uint64(0xffffffff)).astype(np.float64) / 100.0
base = w0.min()
w0 -= base; w1 -= base
print("WALL CLOCK (100 MHz, device-wide): kernel span %.1f us; tile starts: p50 %.1f p90 %.1f p97 %.1f max %.1f; tile ends: p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
    w1.max(), np.median(w0), np.percentile(w0, 90), np.percentile(w0, 97), w0.max(), np.median(w1), np.percentile(w1, 90), np.percentile(w1, 99), w1.max()))
late = np.argsort(-w1)[:10]
print("last tiles by wall clock: blk start end Se S_eff:", [(int(b), round(float(w0[b]), 1), round(float(w1[b]), 1), int(Se[b]), int(Seff[b])) for b in late])
print("tiles starting after 5 us: %d" % int((w0 > 5).sum()))
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
