import os, sys, json
sys.path.insert(0, "/root/repo")
from panovlm_amd import synthetic as sy
from tests import host_io
n = int(sys.argv[1]) if len(sys.argv) > 1 else 454
base = []
for k in range(16):
    R, t = sy.estimated_pose(k)
    base.append(dict(id=k, R_wl=R, t_wl=t, raw=sy.raw_vlp16_scan(k, clutter=40)))
scans = [dict(base[k % 16], id=k) for k in range(n)]
path = "/tmp/raw_%d.bin" % n
host_io.write_raw_scans(path, scans)
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for l in host_io.run("featbench_gpu", path, 3, 1, threads): print(l)
for l in host_io.run("featbench", "/tmp/raw_16.bin" if os.path.exists("/tmp/raw_16.bin") else path, 1, 1): print(l)
