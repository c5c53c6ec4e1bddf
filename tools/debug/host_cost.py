"""Host-side cost of issuing one step (GPU box): wall time of the issue loop against the GPU's."""
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import CFG_B
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
dev = torch.device("cuda:0")
model = LiftRenderStep(CFG_B, dev); batch = SyntheticBatch(CFG_B, 1, dev)
def step():
    model.zero_grad(set_to_none=True); train_step(model, batch)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {(t1 - t0) / 200 * 1e3:.3f} ms/step, until the GPU is done {(t2 - t0) / 200 * 1e3:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
