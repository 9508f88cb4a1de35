#!/usr/bin/env python3
"""The data_vis sweep at its full size on one GPU: 64 lines = 8 contents x (4 character x 2 background styles),
14 SD-v1-sized feature pickles (50 GB) in tmpfs, through the run.py loop (bench.sweep_record with these counts):
imgs/s including the feature load by route, resident feature lists under the LRU bound.
    python tools/sweep64.py > gpurun_out/sweep64.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")        # what run.py --lanes asks for (before the runtime starts)
import torch
import bench
from style_transfer_amd import lib

lib.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench._LANE_STREAMS[:] = [torch.cuda.Stream(device=dev) for _ in range(4)]
sys.argv.append("--progress")
rec = bench.sweep_record(dev, "f16", n_cnt=8, n_char=4, n_back=2, reference_lines=2)
print(json.dumps(rec))
