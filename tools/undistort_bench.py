import sys, time, json
sys.path.insert(0, "/root/repo")
import bench, panovlm_amd as pv
ctx = pv.Context()
print(json.dumps(bench.undistort_block(ctx, pv))[:700])
