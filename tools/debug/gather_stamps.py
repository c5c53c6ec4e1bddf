#!/usr/bin/env python3
"""Debug aid (GPU box): phase stamps of the camera gather's x-run workgroups (wall_clock64, 100 MHz).
Build the diagnostic library first:  tools/ablate.sh render_bwd_cell.hip stamps=-DVAMP_GATHER_STAMPS
Run:  VAMPIRE_HIP_LIB=vampire_amd/_lib/abl_stamps.so python tools/debug/gather_stamps.py"""
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
vols = [v.clone().requires_grad_(True) for v in batch.vols]
lib = _capi.load()
n = 32768
buf = (C.c_longlong * (n * 8))()
lib.vamp_debug_gather_stamps.argtypes = [C.c_void_p, C.c_size_t]
for it in range(4):
    outs = hp.render(*vols, model.beta, render_mats=batch.render_mats)
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
torch.cuda.synchronize()
assert lib.vamp_debug_gather_stamps(buf, n * 8) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(n, 8).astype(np.float64)
seen = a[a[:, 0] > 0]
t0 = seen[:, 0].min()
act = seen[seen[:, 1] >= seen[:, 0].min()]                # passed the flag in THIS launch
act = act[act[:, 1] >= act[:, 0]]
work = act[(act[:, 4] >= act[:, 1]) & (act[:, 2] >= act[:, 1])]
print("workgroups launched %d, flagged %d, with records %d" % (len(seen), len(act), len(work)))
print("launch: first start -> last start %.1f us; last end %.1f us" % ((seen[:, 0].max() - t0) / 100, (work[:, 4].max() - t0) / 100))
st = (work[:, 1] - t0) / 100
print("start of the working workgroups: p10 %.1f median %.1f p90 %.1f max %.1f us" % tuple(np.percentile(st, [10, 50, 90, 100])))
ph = np.stack([work[:, 2] - work[:, 1], work[:, 3] - work[:, 2], work[:, 4] - work[:, 3]], 1) / 100
for i, nm in enumerate(["cell ranges + first values", "record loop", "reduce + store"]):
    print("  %-28s mean %6.2f median %6.2f p90 %6.2f max %6.2f us" % (nm, ph[:, i].mean(), np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max()))
tot = (work[:, 4] - work[:, 1]) / 100
print("  total: mean %.2f median %.2f p90 %.2f max %.2f us; records of thread 0's voxel: mean %.1f max %d" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), work[:, 6].mean(), work[:, 6].max()))
end = (work[:, 4] - t0) / 100
print("ends: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(end, [50, 90, 99, 100])))
