#!/usr/bin/env python3
"""Debug aid (GPU box): wall-clock start / end of every workgroup of the merged render forward (camera tiles, then BEV
column blocks) -- the timeline that shows the BEV blocks running in the camera tiles' tail.
Build the diagnostic library first:  tools/ablate.sh render_fwd_merged.hip mstamps=-DVAMP_MERGED_STAMPS
Run:  VAMPIRE_HIP_LIB=vampire_amd/_lib/abl_mstamps.so python tools/debug/merged_stamps.py [cfg] [batch] [out.txt] [train]
(`train`: the training launch -- ranks drawn, sample rows and BEV planes kept -- instead of the no-grad one)"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch
from vampire_amd import _capi
name = sys.argv[1] if len(sys.argv) > 1 else "B"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = PRESETS[name]
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev)
batch = SyntheticBatch(cfg, B, dev, seed=0)
hp = model.hp
train = len(sys.argv) > 4 and sys.argv[4] == "train"
if train:
    from vampire_amd.step import train_step
    for _ in range(5):
        model.zero_grad(set_to_none=True)
        train_step(model, batch)
    torch.cuda.synchronize()
else:
    with torch.no_grad():
        for _ in range(5):
            hp.render(*batch.vols, model.beta, render_mats=batch.render_mats)
        torch.cuda.synchronize()
lib = _capi.load()
tiles = B * cfg.num_cams * ((cfg.fH + 7) // 8) * ((cfg.fW + 7) // 8)
ncam = (tiles + 7) // 8 * 8
gx = ((cfg.oY * cfg.oX + 63) // 64 + 7) // 8 * 8
nbev = gx * B * 3                      # VAMP_MERGED_BEV_PARTS channel groups per column block
n = min(ncam + nbev, 8192)
buf = (C.c_longlong * (8192 * 2))()
lib.vamp_debug_merged_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.vamp_debug_merged_stamps(buf, 8192 * 2) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(8192, 2)[:n].astype(np.float64) / 100.0     # 100 MHz wall clock -> us
live = a[:, 1] > 0
t0 = a[live, 0].min()
s, e = a[:, 0] - t0, a[:, 1] - t0
cam = np.arange(n) < ncam
# (BEV workgroups past the lattice return at once: they are the grid's padding)
lines = []
def P(x):
    print(x); lines.append(x)
P(f"merged render forward ({'training' if train else 'no-grad'}), cfg-{name} x{B}: {ncam} camera workgroups + {nbev} BEV workgroups; span {e[live].max():.1f} us")
for nm, m in (("camera tiles", cam & live), ("BEV blocks", ~cam & live & ((e - s) > 0.5))):
    P(f"{nm:13s} n={int(m.sum()):5d}  start p0/p50/p90/max {s[m].min():5.1f} {np.median(s[m]):5.1f} {np.percentile(s[m], 90):5.1f} {s[m].max():5.1f}   "
      f"end p50/p90/p99/max {np.median(e[m]):5.1f} {np.percentile(e[m], 90):5.1f} {np.percentile(e[m], 99):5.1f} {e[m].max():5.1f}   "
      f"duration mean/p50/p90/max {(e - s)[m].mean():5.1f} {np.median((e - s)[m]):5.1f} {np.percentile((e - s)[m], 90):5.1f} {(e - s)[m].max():5.1f} us")
P("timeline (workgroups running at t): t us | camera | BEV")
for t in np.arange(0, e[live].max() + 2, 2.0):
    rc = int(((s <= t) & (e > t) & cam & live).sum())
    rb = int(((s <= t) & (e > t) & ~cam & live).sum())
    P(f"  {t:5.1f} | {rc:5d} {'#' * (rc // 32)} | {rb:5d} {'+' * (rb // 32)}")
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(lines) + "\n")
