#!/usr/bin/env python3
"""Debug aid (GPU box): the layered step of bench.py alone (for tools/ktail.sh: steady-state kernel list).
usage: tools/debug/layered_time.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pass  # (bench sets the MIOpen db on import)
import torch
from vampire_amd.config import PRESETS
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
r = bench.layered_measure(PRESETS["B"], dev, 1, 0, 1, steps=n, warm=4, find=True)
print("layered ms/step %.2f" % r["ms_per_step"])
