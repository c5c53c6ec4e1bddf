import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print("MIOPEN_DB", bench.MIOPEN_DB, os.listdir(bench.MIOPEN_DB) if bench.MIOPEN_DB else None)
import torch
print(torch.backends.cudnn.version(), torch.cuda.get_device_properties(0).gcnArchName)
print("matches", bench._miopen_db_matches())
