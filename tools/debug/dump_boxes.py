"""Diagnostic (GPU box): dump the sub-brick boxes of one camera forward (-DVAMP_DUMP_BOXES build)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vampire_amd.config import PRESETS
from vampire_amd.geometry import render_matrices
from vampire_amd import synthetic
from vampire_amd.ops import HotPath
cfg = PRESETS["B"]
dev = torch.device("cuda:0")
hp = HotPath(cfg, dev)
s2e, K, ida = synthetic.camera_rig(cfg, 1)
rm = render_matrices(s2e, K, ida, synthetic.bda_matrix(1)).to(dev)
vols = synthetic.render_inputs(cfg, 1, device=dev)
beta = torch.tensor(0.1, device=dev)
hp.impl["overlap"] = False
raw = C.CDLL(os.environ["VAMPIRE_HIP_LIB"])
buf = (C.c_int * (65536 * 8))(); n = C.c_uint()
with torch.no_grad():
    hp.render(*vols, beta, render_mats=rm); torch.cuda.synchronize()
    raw.vamp_debug_boxes(buf, C.byref(n), 1)
    hp.render(*vols, beta, render_mats=rm); torch.cuda.synchronize()
    raw.vamp_debug_boxes(buf, C.byref(n), 0)
a = np.frombuffer(buf, dtype=np.int32).reshape(-1, 8)[:min(n.value, 65536)].copy()
os.makedirs(os.path.join(ROOT, "gpurun_out", "dbg"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "dbg", "boxes.npy"), a)
print("boxes", n.value, "rows mean", (a[:,3]*a[:,4]*a[:,5]).mean(), "ex mean", a[:,3].mean(), a[:,4].mean(), a[:,5].mean())
