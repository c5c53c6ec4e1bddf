"""Diagnostic (GPU box): phase cycle sums of the brick forward kernel from a -DVAMP_STAMP build."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vampire_amd.config import PRESETS
from vampire_amd.geometry import render_matrices
from vampire_amd import synthetic, _capi
from vampire_amd.ops import HotPath
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
dev = torch.device("cuda:0")
hp = HotPath(cfg, dev)
s2e, K, ida = synthetic.camera_rig(cfg, 1)
rm = render_matrices(s2e, K, ida, synthetic.bda_matrix(1)).to(dev)
vols = synthetic.render_inputs(cfg, 1, device=dev)
beta = torch.tensor(0.1, device=dev)
hp.impl["overlap"] = False
raw = C.CDLL(os.environ["VAMPIRE_HIP_LIB"])
buf = (C.c_ulonglong * 16)()
with torch.no_grad():
    for _ in range(3):
        hp.render(*vols, beta, render_mats=rm)
    torch.cuda.synchronize()
    raw.vamp_debug_stamps(buf, 1)
    n = 5
    for _ in range(n):
        hp.render(*vols, beta, render_mats=rm)
    torch.cuda.synchronize()
    raw.vamp_debug_stamps(buf, 0)
names = ["plan", "skipped idx", "points+taps", "wait+gather", "prefetch issue", "composite", "whole wave"]
waves = buf[7] / n
print(f"waves per launch {waves:.0f}")
for k, nm in enumerate(names):
    print(f"  {nm:16s} {buf[k] / n / waves:10.0f} cycles per wave (s_memtime ticks, 100 MHz?)  total {buf[k]/n:.3e}")
for k, nm in enumerate(["sub-brick gathers", "serial (not prefetched) DMAs", "cycles: dma issue+wait", "cycles: LDS taps", "rows staged", "cycles: dma issue only (incl. pre-waits)"]):
    print(f"  {nm:30s} per launch {buf[8 + k] / n:.4e}  per wave {buf[8 + k] / n / waves:.1f}")
