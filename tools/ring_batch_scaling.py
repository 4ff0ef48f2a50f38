"""HIP-event milliseconds of the range-image stages (pvlm_ring_extract_batch, K16-K23 + copies) against the number of scans in a batch."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panovlm_amd as pv
from panovlm_amd import synthetic as sy

ctx = pv.Context()
base = [sy.raw_vlp16_scan(k, clutter=40) for k in range(16)]
for n in ((454, 454) if "--picks" in sys.argv else (454, 28, 57, 114, 227, 454)):
    raws = [base[k % 16] for k in range(n)]
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        b = pv.RingBatch(ctx, raws, picks=(1000.0, 5.0) if "--picks" in sys.argv else None)
        wall = 1e3 * (time.perf_counter() - t0)
        t = b.timing()
        b.close()
        if best is None or wall < best[0]:
            best = (wall, t)
    print("scans %4d wall_ms %7.2f events_ms %7.2f  " % (n, best[0], sum(best[1].values())) + " ".join("%s %.2f" % (k, v) for k, v in best[1].items()))
