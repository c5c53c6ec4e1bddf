#!/usr/bin/env python3
"""GPU box: run the layered and the multi-task leg of bench.py with MIOpen's find, with the user find-db and the
kernel cache pointed at gpurun_out/miopen_db (copy it to profiles/miopen_db afterwards: bench.py points MIOpen
there, so that a default bench run gets the tuned convolutions without the 6.5 minutes of find).
usage: tools/miopen_tune.py   (environment is set here, before torch is imported)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
db = os.path.join(ROOT, "gpurun_out", "miopen_db")
os.makedirs(db, exist_ok=True)
os.environ["MIOPEN_USER_DB_PATH"] = db
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = db
sys.path.insert(0, ROOT)
import torch
import bench
from vampire_amd.config import PRESETS
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
t0 = time.time()
print("layered", bench.layered_measure(PRESETS["B"], dev, 1, 0, 1, find=True)["ms_per_step"], "ms", time.time() - t0, "s", flush=True)
t0 = time.time()
print("multitask", bench.multitask_measure(dev, 1, 0, 1, find=True)["ms_per_step"], "ms", time.time() - t0, "s", flush=True)
for f in sorted(os.listdir(db)):
    print(f, os.path.getsize(os.path.join(db, f)))
