import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
cfg = PRESETS["B"]; dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev); batch = SyntheticBatch(cfg, 1, dev)
print("fresh:", bench.forward_pair_us(model, batch))
for _ in range(10):
    model.zero_grad(set_to_none=True); train_step(model, batch)
torch.cuda.synchronize()
print("after training steps:", bench.forward_pair_us(model, batch))
model.hp.impl["overlap"] = False
print("one stream:", bench.forward_pair_us(model, batch))
import cProfile, pstats
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
    torch.cuda.synchronize(); pr.disable()
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(200):
        model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
torch.cuda.synchronize()
print("200 forwards back to back (queue ahead): %.1f us each" % ((time.perf_counter() - t0) / 200 * 1e6))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
