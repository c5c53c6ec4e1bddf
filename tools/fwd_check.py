#!/usr/bin/env python3
"""Development aid (GPU box): forward of lift + render at a full-size config against the committed
reference checksums (value statistics only), plus HIP-event timings of lift / render / the pair."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vampire_amd.config import PRESETS
from vampire_amd import synthetic, _capi
from vampire_amd.ops import HotPath

NAMES = ["rgb_preds", "seg_logits_preds", "depth_preds", "bev_rgb_preds",
         "bev_seg_logits_preds", "bev_height_preds", "voxel_density", "voxel_output"]

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="B")
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
cfg = PRESETS[a.cfg]
dev = torch.device("cuda:0")
ref = json.load(open(os.path.join(ROOT, "tests/golden/full_checksums.json")))[a.cfg]
hp = HotPath(cfg, dev)
lm = torch.tensor(ref["lift_mats"], dtype=torch.float32, device=dev)
rm = torch.tensor(ref["render_mats"], dtype=torch.float32, device=dev)
depth, feat = synthetic.lift_inputs(cfg, 1, seed=0, device=dev)
vols = synthetic.render_inputs(cfg, 1, seed=0, device=dev)
beta = torch.tensor(0.1, device=dev)
with torch.no_grad():
    vox = hp.lift(depth, feat, lm)
    outs = hp.render(*vols, beta, render_mats=rm)
torch.cuda.synchronize()
ok = True
def chk(name, t, st, probe=None):
    global ok
    tot = float(t.double().abs().sum())
    e = abs(tot - st["abs_sum"]) / max(st["abs_sum"], 1e-9)
    flag = "ok" if e <= 2e-5 else "MISMATCH"
    if flag != "ok": ok = False
    print(f"  {name:24s} abs_sum rel err {e:.2e} {flag}")
chk("lift", vox, ref["lift"])
for n, o in zip(NAMES, outs):
    chk(n, o, ref[n])
p = outs[2].flatten()[::1013][:64].cpu()
print("  depth probe max err", float((p - torch.tensor(ref["depth_preds_probe"])).abs().max()))
p = outs[1].flatten()[::10007][:64].cpu()
print("  seg probe max err", float((p - torch.tensor(ref["seg_probe"])).abs().max()))

def timeit(fn, iters=a.iters, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
with torch.no_grad():
    tl = timeit(lambda: hp.lift(depth, feat, lm))
    tr = timeit(lambda: hp.render(*vols, beta, render_mats=rm))
    tp = timeit(lambda: (hp.lift(depth, feat, lm), hp.render(*vols, beta, render_mats=rm)))
    _capi.profile_select(None); _capi.profile_enable(True)
    for _ in range(5):
        hp.lift(depth, feat, lm); hp.render(*vols, beta, render_mats=rm)
    torch.cuda.synchronize()
    _capi.profile_enable(False)
ab = cfg.algorithmic_bytes()
print(f"cfg {a.cfg}: lift {tl:.1f} us, render {tr:.1f} us, pair {tp:.1f} us -> {ab['fwd']/tp/1e6/8*100:.1f}% of 8 TB/s (median of {a.iters})")
for k, (n, ms) in sorted(_capi.profile_read().items()):
    print(f"  {k:28s} {ms / n * 1e3:8.1f} us x{n}")
print("PARITY", "OK" if ok else "FAILED")

# ---- the same pair replayed from a HIP graph (no per-kernel host launch cost)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.no_grad():
    with torch.cuda.stream(side):
        for _ in range(3):
            hp.lift(depth, feat, lm); hp.render(*vols, beta, render_mats=rm)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        hp.lift(depth, feat, lm); hp.render(*vols, beta, render_mats=rm)
    tg = timeit(lambda: graph.replay())
    hp.impl["overlap"] = False
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        hp.lift(depth, feat, lm); hp.render(*vols, beta, render_mats=rm)
    torch.cuda.synchronize()
    with torch.cuda.graph(g2, stream=side):
        hp.lift(depth, feat, lm); hp.render(*vols, beta, render_mats=rm)
    tg2 = timeit(lambda: g2.replay())
    te = timeit(lambda: (hp.lift(depth, feat, lm), hp.render(*vols, beta, render_mats=rm)))
print(f"graph replay of the pair: two streams {tg:.1f} us ({ab['fwd']/tg/1e6/8*100:.1f}% of 8 TB/s), one stream {tg2:.1f} us; eager one stream {te:.1f} us")
