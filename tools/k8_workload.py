#!/usr/bin/env python3
"""The camera<->LiDAR voting block of bench.py on its own (1 362 pairs x 200 image lines x ~1 500 corner points), for rocprofv3 passes and A/B runs:
prints the `cam_lidar_votes` object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import panovlm_amd as pv
ctx = pv.Context(0)
out = bench.panorama_block(ctx, pv, torch, torch.device("cuda:0"), with_votes=True)
print(json.dumps(out["cam_lidar_votes"]))
