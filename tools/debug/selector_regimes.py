"""Debug aid (GPU box): the data-chosen camera forward in one density regime -- which forward the calls take and what a
step costs.  usage: tools/debug/selector_regimes.py [naive|sdf]"""
import os, sys, time, dataclasses, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import PRESETS
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
mode = sys.argv[1] if len(sys.argv) > 1 else "naive"
cfg = dataclasses.replace(PRESETS["B"], density_mode=mode)
dev = torch.device("cuda:0")
model = LiftRenderStep(cfg, dev)
data = SyntheticBatch(cfg, 1, dev, seed=1)
hp = model.hp
def t(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for i in range(6):
    train_step(model, data); torch.cuda.synchronize()
    print(i, hp._cam_sel["mode"], float(hp._cam_sel["host"][0]), float(hp._cam_sel["dev"][0]))
print("train step us", t(lambda: train_step(model, data)))
with torch.no_grad():
    f = lambda: model(data.depth, data.feat, data.vols, data.lift_mats, data.render_mats)
    print("fwd us", t(f), hp._cam_sel)
for forced in (True, False):
    hp.impl["cam_direct"] = forced
    for e in (True, False):
        hp.impl["ert"] = e
        print("direct", forced, "ert", e, "train", t(lambda: train_step(model, data)), end=" ")
        with torch.no_grad():
            print("fwd", t(f))
