#!/usr/bin/env python3
"""The no-grad forward pair (bench.py's fwd_roofline protocol) captured in a HIP graph and replayed.
usage: tools/fwd_graph.py [cfg] [batch] [replays] [overlap 0|1] [impl overrides: key=0|1 ...]
Under `rocprofv3 --kernel-trace` + tools/debug/graph_timeline.py it gives the kernel timeline of a replay."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
overlap = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev)
over = dict(kv.split("=") for kv in sys.argv[5:])
batch = SyntheticBatch(cfg, B, dev, feat_channel_last=over.pop("feat_cl", "1") != "0")     # (feat_cl=0: [B,N,C,fH,fW] features)
hp = model.hp
for k, v in over.items():
    hp.impl[k] = {"0": False, "1": True}.get(v, v)
with torch.no_grad():
    def fwd():
        return model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
    fwd()
    hp.impl["fwd_overlap"] = overlap
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        fwd()
    torch.cuda.synchronize()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record(); b.synchronize()
    b2b = a.elapsed_time(b) * 1e3 / n
print(f"forward pair replayed (overlap={overlap} {' '.join(sys.argv[5:])}): back to back {b2b:.1f} us; median {ts[len(ts)//2]:.1f} us, p10 {ts[len(ts)//10]:.1f}, p90 {ts[9*len(ts)//10]:.1f}")
