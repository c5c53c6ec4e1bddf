#!/usr/bin/env python3
"""Cost of the DDP wrapper on the 1 ms step (what the N > 1 bench adds per rank besides the wire
time): a world-size-1 RCCL process group on one GPU, DDP(LiftRenderStep) against the bare module."""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import CFG_B
from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1, device_id=dev)
model = LiftRenderStep(CFG_B, dev)
batch = SyntheticBatch(CFG_B, 1, dev)
ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
from vampire_amd.dist import GradSync
model2 = LiftRenderStep(CFG_B, dev)
hook = GradSync(model2)


def timed(m, n=300):
    for _ in range(20):
        m.zero_grad(set_to_none=True); train_step(m, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m.zero_grad(set_to_none=True); train_step(m, batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, m in (("bare    ", model), ("DDP     ", ddp), ("GradSync", hook), ("bare    ", model), ("DDP     ", ddp),
                ("GradSync", hook)):
    print("%s  %.4f ms/step" % (name, timed(m)), flush=True)
dist.destroy_process_group()
