#!/usr/bin/env python3
"""Per-kernel times of the renderer (GPU box; development aid): the library's profiler slots (a HIP-event
pair around every launch, on its own stream) over N one-stream calls -- forward-only and training.

  python tools/time_render.py [--cfg B] [--batch 1] [--iters 30] [--only render_bev]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd import _capi, synthetic
from vampire_amd.config import PRESETS
from vampire_amd.geometry import render_matrices
from vampire_amd.ops import HotPath


def slots(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _capi.profile_select(None)
    _capi.profile_read()
    _capi.profile_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    _capi.profile_enable(False)
    return {k: (n / iters, ms / max(n, 1) * 1e3) for k, (n, ms) in _capi.profile_read().items() if n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="B")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--impl", default="", help="HotPath.impl overrides, e.g. ert=0,cam_direct=0,overlap=1")
    a = ap.parse_args()
    cfg = PRESETS[a.cfg]
    dev = torch.device("cuda:0")
    hp = HotPath(cfg, dev)
    hp.impl["overlap"] = False
    for kv in filter(None, a.impl.split(",")):
        k, v = kv.split("=")
        hp.impl[k] = bool(int(v)) if v in ("0", "1") else (int(v) if v.lstrip("-").isdigit() else v)
    B = a.batch
    s2e, K, ida = synthetic.camera_rig(cfg, B)
    rm = render_matrices(s2e, K, ida, synthetic.bda_matrix(B)).to(dev)
    vols = synthetic.render_inputs(cfg, B, device=dev)
    beta = torch.tensor(0.1, device=dev)

    def fwd():
        with torch.no_grad():
            hp.render(*vols, beta, render_mats=rm)

    tv = [v.clone().requires_grad_(True) for v in vols]
    tb = beta.clone().requires_grad_(True)
    outs = hp.render(*tv, tb, render_mats=rm)
    gs = [torch.randn_like(o) for o in outs]

    def train():
        o = hp.render(*tv, tb, render_mats=rm)
        torch.autograd.backward(o, gs)

    for name, fn in (("forward only", fwd), ("training", train)):
        r = slots(fn, a.iters)
        print(f"[{name}] cfg-{a.cfg} B={B}")
        for k, (n, us) in sorted(r.items()):
            if a.only and a.only not in k:
                continue
            print(f"  {k:28s} {n:4.1f} launches/call  {us:8.2f} us")


if __name__ == "__main__":
    main()
