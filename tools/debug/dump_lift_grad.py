"""Debug aid: dump the HIP lift gradients at cfg-A/B (reference-fixture inputs) to gpurun_out/."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vampire_amd.config import CFG_A, CFG_B
from vampire_amd import synthetic
from vampire_amd.ops import HotPath
name = sys.argv[1]
impl = sys.argv[2] if len(sys.argv) > 2 else "cell"
cfg = {"A": CFG_A, "B": CFG_B}[name]
dev = torch.device("cuda:0")
ref = json.load(open(os.path.join(ROOT, "tests/golden/full_grad_checksums.json")))[name]
mats = json.load(open(os.path.join(ROOT, "tests/golden/full_checksums.json")))[name]
hp = HotPath(cfg, dev)
hp.impl["lift_bwd"] = impl
lm = torch.tensor(mats["lift_mats"], dtype=torch.float32, device=dev)
depth, feat = synthetic.lift_inputs(cfg, 1, seed=0, device=dev)
depth.requires_grad_(True); feat.requires_grad_(True)
vox = hp.lift(depth, feat, lm)
g = torch.Generator().manual_seed(ref["seed_lift"])
up = (torch.randn(vox.shape, generator=g) * 1e-3).to(dev)
vox.backward(up)
out = os.path.join(ROOT, "gpurun_out", "dbg")
os.makedirs(out, exist_ok=True)
np.save(os.path.join(out, f"hip_gdepth_{name}_{impl}.npy"), depth.grad.cpu().numpy())
np.save(os.path.join(out, f"hip_gfeat_{name}_{impl}.npy"), feat.grad.cpu().numpy())
print("saved")
