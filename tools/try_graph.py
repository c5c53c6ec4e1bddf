#!/usr/bin/env python3
"""Experiment: one training step of the hot path captured in a HIP graph (torch.cuda.CUDAGraph)
against the eager step.  usage: tools/try_graph.py [cfg] [batch] [steps] [HotPath.impl overrides: key=0|1 ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step

cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "B"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev)
batch = SyntheticBatch(cfg, B, dev)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    model.hp.impl[k] = {"0": False, "1": True}.get(v, int(v) if v.isdigit() else v)


def step():
    model.zero_grad(set_to_none=True)
    return train_step(model, batch)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(20):
    step()
print("eager  %.4f ms/step" % timed(step, steps), flush=True)
ref = [t.detach().clone() for t in (batch.depth.grad, batch.feat.grad, *[v.grad for v in batch.vols], model.beta.grad)]

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
got = [batch.depth.grad, batch.feat.grad, *[v.grad for v in batch.vols], model.beta.grad]
for a, b in zip(got, ref):
    assert torch.equal(a, b) or float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()), "graph replay differs"
print("graph  %.4f ms/step (replay == eager gradients)" % timed(g.replay, steps), flush=True)
if os.environ.get("TRY_EAGER_LAST") == "1":          # (timeline runs: make the trace end with an eager step)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
