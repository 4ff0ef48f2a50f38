"""BASELINE.json config 4 on the product path, on ONE GPU: LidarOdometry::EstimatePose at the reference's Floor size (config/Floor.txt:
1593 scans, lidar_plane_tolerance 0.01, point-to-plane + line-to-line) as one process and sharded over 2 and over 8 processes that
share the GPU (each rank its own context, HIP kernels for association and normal equations, the per-iteration sum of the packed
blocks through a directory — RCCL refuses several ranks on one device).  The loop that is sharded: util/Optimization.cpp:521-560 and
:345-441 under lidar_mapping/LidarOdometry.cpp:116-187.  Every rank must report the same bits; the sharded poses must equal the
one-process poses to 1e-9 with identical step and block counts; the partition must be balanced (incl. the
loop-closure pairs the bouncing trajectory produces)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_estimate_pose_at_floor_scale_sharded_over_2_and_8_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "floor_like_odometry.py"), "--scans", "1593", "--ranks", "2,8", "--iters", "2"],
                       capture_output=True, text=True, timeout=2400, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = r.stdout
    assert "sharded runs equal to the one-process run (1e-9, identical step counts): True" in out, out[-3000:]
    assert out.count("every rank reports the same log and poses, bit for bit: True") == 2
    iters = [l.split() for l in out.splitlines() if l.strip().startswith("iter")]
    assert len(iters) >= 2 and all(int(l[6]) > 500_000 for l in iters)          # the one-process log; Floor tolerance: > 0.5 M residual blocks per outer iteration
    m = re.search(r"mean translation error vs ground truth: ([0-9.]+) m -> ([0-9.]+) m", out)
    assert m and float(m.group(2)) < 0.5 * float(m.group(1))
    bal = [(float(a), float(b)) for a, b in re.findall(r"load balance \(max / mean\): queries ([0-9.]+), residual blocks ([0-9.]+)", out)]
    assert len(bal) == 2 and all(q <= 1.05 and b <= 1.25 for q, b in bal), bal
    ranks8 = re.findall(r"rank (\d): reference scans \[(\d+), (\d+)\)", out)
    assert len(ranks8) == 2 + 8 and ranks8[-1][2] == "1593"
