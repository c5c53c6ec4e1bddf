"""Top ops of the layered step (step.LayeredStep, cfg-B, fp32): torch profiler."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vampire_amd.config import CFG_B
from vampire_amd.step import LayeredStep, LayeredBatch, layered_step
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
model = LayeredStep(CFG_B, dev)
data = LayeredBatch(CFG_B, 1, dev, seed=0)
for _ in range(4):
    model.zero_grad(set_to_none=True); layered_step(model, data)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    model.zero_grad(set_to_none=True); layered_step(model, data)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
