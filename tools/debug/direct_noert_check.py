#!/usr/bin/env python3
"""Debug aid (GPU box): cfg-B, sdf, cat_seg, early termination off -- the volume gradients of the default backward
behind the one-kernel camera forward, behind copy + march, and of the v1 float-atomic backward, against each other."""
import dataclasses, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import CFG_B
from vampire_amd.geometry import render_matrices
from vampire_amd import synthetic
from vampire_amd.ops import HotPath

dev = torch.device("cuda:0")
cfg = dataclasses.replace(CFG_B, density_mode="sdf", cat_seg=True)
hp = HotPath(cfg, dev)
hp.impl["ert"] = False
s2e, K, ida = synthetic.camera_rig(cfg, 1, jitter=2.0, seed=7)
rm = render_matrices(s2e, K, ida, synthetic.bda_matrix(1, rot_deg=-4.0)).to(dev)
beta = torch.tensor(0.1, device=dev, requires_grad=True)
gen = torch.Generator(device=dev)


def run(direct, impl, **kw):
    hp.impl["cam_direct"] = direct
    hp.impl["cam_bwd"] = hp.impl["bev_bwd"] = impl
    for k, v in kw.items():
        hp.impl[k] = v
    vols = [v.requires_grad_(True) for v in synthetic.render_inputs(cfg, 1, seed=6, device=dev)]
    beta.grad = None
    outs = hp.render(*vols, beta, render_mats=rm)
    gen.manual_seed(3)
    gs = [torch.randn(o.shape, device=dev, generator=gen) for o in outs]
    torch.autograd.backward(outs, gs)
    torch.cuda.synchronize()
    return [o.detach().clone() for o in outs], [v.grad.clone() for v in vols]


ref_o, ref_g = run(False, "v1")
names = ("density_feature", "semantic_logits", "base", "rgb")
for tag, args, kw in (("copy+march, cell", (False, "cell"), {}), ("one kernel, cell", (True, "cell"), {}),
                      ("one kernel, cell, no kept rows", (True, "cell"), {"save_rows": False}),
                      ("one kernel, cell, one stream", (True, "cell"), {"save_rows": True, "overlap": False}),
                      ("one kernel, v1", (True, "v1"), {"overlap": True})):
    o, g = run(*args, **kw)
    errs = ["%s %.2e" % (n, float((a - b).abs().max() / b.abs().max())) for n, a, b in zip(names, g, ref_g)]
    oerr = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(o, ref_o))
    print("%-34s outputs %.2e | grads %s" % (tag, oerr, "  ".join(errs)))
