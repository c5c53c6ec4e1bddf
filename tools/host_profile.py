#!/usr/bin/env python3
"""cProfile of the host side of one training step (dev aid)."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
cfg = PRESETS["B"]; dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev); batch = SyntheticBatch(cfg, 1, dev)
for _ in range(5): train_step(model, batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): train_step(model, batch)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/50:.3f} ms/step, total {1e3*(t2-t0)/50:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): train_step(model, batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
