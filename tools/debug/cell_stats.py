#!/usr/bin/env python3
"""GPU box: how the camera backward's records distribute over cells and voxels (cfg-B sample of the bench; early termination on, or -- argument `noert` -- off)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch
cfg = PRESETS["B"]; dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev); batch = SyntheticBatch(cfg, 1, dev); hp = model.hp
with torch.no_grad():
    hp.render(*batch.vols, model.beta, render_mats=batch.render_mats)
d = hp.render_desc(1, cfg.num_cams, 0)
off = hp.lib.vamp_render_term_offset(C.byref(d)); n = cfg.num_cams * cfg.fH * cfg.fW
term = hp._ws["render"][off:off + 4 * n].view(torch.int32).reshape(1, cfg.num_cams, 1, cfg.fH, cfg.fW).clone()
inside, ix0, iy0, iz0 = hp.render_indices(render_mats=batch.render_mats)
idx = torch.arange(cfg.D - 1, device=dev).reshape(1, 1, -1, 1, 1)
kept = inside.bool() & (idx < term)
if "noert" in sys.argv[1:]:                # every inside sample (early termination off)
    kept = inside.bool()
print("inside", int(inside.sum()), "kept", int(kept.sum()))
X, Y, Z = cfg.vX, cfg.vY, cfg.vZ
cell = ((iz0.long() + 1) * (Y + 1) + (iy0.long() + 1)) * (X + 1) + (ix0.long() + 1)
cnt = torch.bincount(cell[kept], minlength=(Z + 1) * (Y + 1) * (X + 1)).reshape(Z + 1, Y + 1, X + 1)
print("cells with records", int((cnt > 0).sum()), "max per cell", int(cnt.max()))
for T in (4, 8, 16, 32, 64, 128, 256, 512):
    m = cnt > T
    print(f"  cells > {T}: {int(m.sum())} holding {int(cnt[m].sum())} records ({100.0 * float(cnt[m].sum()) / float(cnt.sum()):.1f} %)")
# records per voxel = sum over its 8 cells
v = sum(cnt[dz:dz + Z, dy:dy + Y, dx:dx + X] for dz in (0, 1) for dy in (0, 1) for dx in (0, 1))
print("voxels with records", int((v > 0).sum()), "max per voxel", int(v.max()), "sum of record visits", int(v.sum()))
for T in (256,):
    m = v > T
    print(f"  voxels > {T}: {int(m.sum())} visiting {int(v[m].sum())} records ({100.0 * float(v[m].sum()) / float(v.sum()):.1f} % of visits)")
# per depth index: kept samples
per_i = kept.sum(dim=(0, 1, 3, 4))
print("kept per depth index (first 16):", per_i[:16].tolist())
