import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
bench._use_shipped_miopen_db() if hasattr(bench, "_use_shipped_miopen_db") else None
for i in range(3):
    r = bench.multitask_measure(dev, 1, 0, 1, steps=6, warm=3)
    print("multitask ms/step %.1f" % r["ms_per_step"], flush=True)
