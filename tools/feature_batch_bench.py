"""Velodyne::ExtractFeaturesBatch (host mirror) over a Room-sized batch of raw VLP-16 scans: wall per call, thread-milliseconds of the host stages
(PVLM_FEATURE_PROFILE) and the A/B switches of the call —
    PVLM_FEATURE_PICKS=host   the picks and the voxel grid on the host threads (PickFeatures) instead of K24
    PVLM_FEATURE_PARTS=k      the batch cut into k device batches whose GPU stages overlap the host work of the previous one (default 2)
    PVLM_EDGE_GROW=host       the growth of the line segments on the host threads instead of K27 (--grow-ab: both, with the default picks / parts)
usage: feature_batch_bench.py [scans = 454] [threads = 32] [--ab | --parts=2,3,4]     (--ab: runs the combinations)
The boxes of this pool give a process 16 CPUs' worth of time per 100 ms (cgroup cpu.max) whatever nproc says: a 32-thread call of ~50 ms fits one period's
budget, two calls back to back do not — the driver sleeps before each repetition, and thread_ms (CPU time actually spent) is the figure that transfers."""
import os, sys
sys.path.insert(0, "/root/repo")
from panovlm_amd import synthetic as sy
from tests import host_io

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 454
threads = int(args[1]) if len(args) > 1 else 32
base = []
for k in range(16):
    R, t = sy.estimated_pose(k)
    base.append(dict(id=k, R_wl=R, t_wl=t, raw=sy.raw_vlp16_scan(k, clutter=40)))
scans = [dict(base[k % 16], id=k) for k in range(n)]
path = "/tmp/raw_%d.bin" % n
host_io.write_raw_scans(path, scans)
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), "hardware threads", os.cpu_count())
except Exception:
    pass
os.environ["PVLM_FEATURE_PROFILE"] = "1"
combos = [("device", "2")] if "--ab" not in sys.argv else [("host", "1"), ("host", "2"), ("device", "1"), ("device", "2"), ("device", "4")]
for a in sys.argv[1:]:
    if a.startswith("--parts="):                      # --parts=2,3,4: the picks on the device with these numbers of device batches per call
        combos = [("device", k) for k in a[8:].split(",")]
for picks, parts in combos:
    os.environ["PVLM_FEATURE_PICKS"] = picks; os.environ["PVLM_FEATURE_PARTS"] = parts
    # PVLM_EDGE_GROW=host: the line segments grown on the host threads (round 5) instead of K27 (pvlm_line_grow_batch)
    for grow in (["gpu", "host"] if picks == "device" and ("--ab" in sys.argv or "--grow-ab" in sys.argv) else ["gpu"]):
        os.environ.pop("PVLM_EDGE_GROW", None)
        if grow == "host": os.environ["PVLM_EDGE_GROW"] = "host"
        print("== picks on the %s, %s device batch(es) per call, %d host threads, line growth on the %s" % (picks, parts, threads, grow if picks == "device" else "host"))
        for l in host_io.run("featbench_gpu", path, 3, 1, threads): print(l)
os.environ.pop("PVLM_EDGE_GROW", None)
for l in host_io.run("featbench", "/tmp/raw_16.bin" if os.path.exists("/tmp/raw_16.bin") else path, 1, 1): print(l)
