"""A world-size-1 RCCL group alive while the step is captured and replayed, all-reduce after each
replay: what bench.py does on N > 1 ranks, on the one GPU of the box."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import bench
from vampire_amd.config import CFG_B
from vampire_amd import dist as vdist
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
vdist.init("nccl", dev, force=True)
model = LiftRenderStep(CFG_B, dev)
sm = vdist.wrap_ddp(model, dev, force=True)
batch = SyntheticBatch(CFG_B, 1, dev)
for _ in range(3):
    model.zero_grad(set_to_none=True); train_step(sm, batch)
torch.cuda.synchronize()
sm.enabled = False
g = bench.capture_step(model, batch, train_step)
sm.enabled = True
for _ in range(20):
    g.replay(); dist.all_reduce(model.beta.grad); model.beta.grad.div_(1)
torch.cuda.synchronize()
print("graph + rccl OK, grad_beta", float(model.beta.grad))
vdist.shutdown()
