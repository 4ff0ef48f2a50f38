#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (pvlm_resset_upload + pvlm_eval with r/J copied
back): what a caller that keeps nothing on the device sees.  Reported in DESIGN.md, never bench `value`."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import panovlm_amd as pv
from tests import synth

rng = np.random.default_rng(0)
F, P, per = 16, 64, 65536
aa, t = synth.random_poses(rng, F)
ref, nei = synth.random_pairs(rng, F, P)
n = P * per
rows = np.empty((n, 7)); rows[:, :3] = rng.normal(size=(n, 3)) * 3
nrm = rng.normal(size=(n, 3)); rows[:, 3:6] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True); rows[:, 6] = rng.normal(size=n)
off = np.arange(P + 1, dtype=np.int64) * per
ctx = pv.Context(0)
t0 = time.perf_counter(); rs = pv.ResidualSet.upload(ctx, 1, rows, off, ref, nei, flags=1); ctx.synchronize(); t_up = time.perf_counter() - t0
ctx.set_poses(aa, t)
rs.eval(jac=True)
t0 = time.perf_counter(); rs.eval(jac=True); t_ev = time.perf_counter() - t0
t0 = time.perf_counter(); rs.eval(jac=False); t_r = time.perf_counter() - t0
print("rows %d: upload %.3f s (%.2f GB/s), eval r+J to host %.3f s = %.1f M evals/s (%.2f GB/s D2H), cost-only %.3f s = %.1f M evals/s" % (
    n, t_up, n * 56 / t_up / 1e9, t_ev, n / t_ev / 1e6, n * 104 / t_ev / 1e9, t_r, n / t_r / 1e6))
