#!/usr/bin/env python3
"""Quick per-operator timing on the GPU box (development aid, not the bench contract)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd.geometry import lift_matrices, render_matrices
from vampire_amd import synthetic
from vampire_amd.ops import HotPath


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="A")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--bwd", action="store_true")
    a = ap.parse_args()
    cfg = PRESETS[a.cfg]
    dev = torch.device("cuda:0")
    hp = HotPath(cfg, dev)
    B = a.batch
    s2e, K, ida = synthetic.camera_rig(cfg, B)
    bda = synthetic.bda_matrix(B)
    lm = lift_matrices(s2e, K, ida, bda).to(dev)
    rm = render_matrices(s2e, K, ida, bda).to(dev)
    depth, feat = synthetic.lift_inputs(cfg, B, device=dev)
    vols = synthetic.render_inputs(cfg, B, device=dev)
    beta = torch.tensor(0.1, device=dev)
    ab = cfg.algorithmic_bytes()
    with torch.no_grad():
        t_lift = timeit(lambda: hp.lift(depth, feat, lm))
        t_render = timeit(lambda: hp.render(*vols, beta, render_mats=rm))
    print(f"cfg {a.cfg} B={B}: lift fwd {t_lift:.1f} us ({ab['lift_fwd']*B/t_lift/1e6:.2f} TB/s alg), "
          f"render fwd {t_render:.1f} us ({ab['render_fwd']*B/t_render/1e6:.2f} TB/s alg), "
          f"fused fwd {(t_lift+t_render):.1f} us -> {ab['fwd']*B/(t_lift+t_render)/1e6/8*100:.1f}% of 8 TB/s")
    if a.bwd:
        depth.requires_grad_(True); feat.requires_grad_(True)
        vols = [v.requires_grad_(True) for v in vols]
        beta.requires_grad_(True)

        def lift_fb():
            o = hp.lift(depth, feat, lm)
            o.backward(torch.ones_like(o))
        def render_fb():
            o = hp.render(*vols, beta, render_mats=rm)
            torch.autograd.backward(o, [torch.ones_like(t) for t in o])
        print(f"  lift fwd+bwd {timeit(lift_fb):.1f} us, render fwd+bwd {timeit(render_fb):.1f} us")


if __name__ == "__main__":
    main()
