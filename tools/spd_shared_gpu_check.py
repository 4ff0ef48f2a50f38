#!/usr/bin/env python3
"""The Floor-shaped pose system of tools/spd_levels_bench.py solved `reps` times by each of `procs` processes that SHARE this GPU: a hash of every solution and its info.
One process: one hash.  Several: the one-launch factorisation (k_nd_flow) may not get through when the processes fight for the GPU's workgroup slots — such a solve is
redone with the level launches by the library (stderr says so), the context keeps them, and a second hash (the level launches' last bits) appears; info stays 0.
python tools/spd_shared_gpu_check.py <reps> <procs>"""
import os, sys, subprocess, hashlib, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
def child(reps):
    from tools.spd_floor_bench import neighbours
    F = 1593
    rng = np.random.default_rng(3)
    pairs = set()
    for i, nb in enumerate(neighbours(F)):
        for j in nb:
            if i != j: pairs.add((min(i, j), max(i, j)))
    pairs = [(p, p) for p in range(F)] + sorted(pairs)
    n = 6 * (F - 1)
    off = np.arange(-6, n).reshape(F, 6); off[0] = -1
    rows = np.array([off[pa] for pa, pb in pairs], np.int32); cols = np.array([off[pb] for pa, pb in pairs], np.int32)
    mirror = np.array([int(pa != pb) for pa, pb in pairs], np.int32)
    blocks = np.empty((len(pairs), 36))
    for k, (pa, pb) in enumerate(pairs):
        if pa == pb:
            J = rng.normal(size=(9, 6)); blocks[k] = (J.T @ J + 30 * np.eye(6)).reshape(-1)
        else:
            blocks[k] = (rng.normal(size=(6, 6)) * 0.2).reshape(-1)
    scale = np.full(n, 0.2); diag = np.full(n, 1.0); rhs = rng.normal(size=n)
    import panovlm_amd as pv
    ctx = pv.Context(0)
    out = []
    for _ in range(reps):
        x, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
        out.append((hashlib.sha1(x.tobytes()).hexdigest()[:10], int(info)))
    print(json.dumps(out))
if len(sys.argv) > 2 and sys.argv[1] == "child":
    child(int(sys.argv[2]))
else:
    procs = [subprocess.Popen([sys.executable, __file__, "child", sys.argv[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(int(sys.argv[2]))]
    res = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        if p.returncode != 0: print("child failed", e[-500:]); continue
        if "ran into its limit" in e: print(e[-6000:])
        res.append(json.loads(o.strip().splitlines()[-1]))
    from collections import Counter
    c = Counter(h for r in res for h, i in r); infos = Counter(i for r in res for h, i in r)
    print("hashes", dict(c), "infos", dict(infos))
