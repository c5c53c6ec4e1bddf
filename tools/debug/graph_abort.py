import os, sys, dataclasses, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vampire_amd.config import CFG_TINY
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
dev = torch.device("cuda:0")
cfg = CFG_TINY
model = LiftRenderStep(cfg, dev); data = SyntheticBatch(cfg, 2, dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
def step():
    model.zero_grad(set_to_none=True)
    data.zero_grads()
    if mode == "lift":
        vox = model.hp.lift(data.depth, data.feat, data.lift_mats); vox.sum().backward()
    elif mode == "render":
        outs = model.hp.render(*data.vols, model.beta, render_mats=data.render_mats)
        sum(o.sum() for o in outs).backward()
    elif mode == "render_cam":
        outs = model.hp.render(*data.vols, model.beta, render_mats=data.render_mats)
        sum(o.sum() for o in outs[:3]).backward()
    else:
        train_step(model, data)
for _ in range(3): step()
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side): step()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize(); print("replayed OK", mode, flush=True)
for it in range(4):
    g.replay(); torch.cuda.synchronize(); print("  replay", it, "OK", flush=True)
which = sys.argv[2] if len(sys.argv) > 2 else "depth"
t = {"depth": data.depth.grad, "feat": data.feat.grad, "dens": data.vols[0].grad, "sem": data.vols[1].grad,
     "base": data.vols[2].grad, "rgb": data.vols[3].grad, "beta": model.beta.grad}[which]
var = sys.argv[3] if len(sys.argv) > 3 else "zero"
print("variant", var, flush=True)
if var == "zero":
    t.zero_(); torch.cuda.synchronize()
elif var == "unrelated":
    u = torch.zeros(1024, device=dev); u.add_(1); torch.cuda.synchronize()
elif var == "sidezero":
    with torch.cuda.stream(side):
        t.zero_()
    torch.cuda.synchronize()
elif var == "hostcopy":
    t.copy_(torch.zeros_like(t, device="cpu")); torch.cuda.synchronize()
if var == "sidereplay":
    t.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(side):
        g.replay()
else:
    g.replay()
torch.cuda.synchronize(); print("  replay after", var, "OK", flush=True)
