"""pvlm_undistort_batch (K25, Velodyne::UndistortCloud of a Room-sized batch) as bench.py's `features.undistort` block measures it: wall per in-place call,
host-link roof, the oracle's per-point loop beside it.  One JSON line."""
import json
import sys
sys.path.insert(0, "/root/repo")
import bench
import panovlm_amd as pv

ctx = pv.Context()
print(json.dumps(bench.undistort_block(ctx, pv)))
