#!/usr/bin/env python3
"""bench.py — M residual+Jacobian evals/s of the point-to-plane hot path on MI355X.

One *step* = one Gauss-Newton/LM linearisation pass over the whole batch of scan pairs at a fresh
parameter point: upload the pose table, evaluate every point-to-plane residual + its 1x12 Jacobian
(the Room/Floor functor Point2Plane_Angle with normalize_distance, HuberLoss(2 deg)), contract them
into the per-pose 6x6 / 6x1 normal-equation blocks, and (N > 1) all-reduce the packed block buffer
over RCCL.  Inputs (correspondence SoA) are resident in HBM before the timed region.

Contract: `python bench.py --gpus N --steps K --warmup W` (torchrun launches N ranks); rank 0
prints ONE JSON line.  Scaling is STRONG: the batch (`--scans` x `--neighbors` ordered pairs) is
fixed and sharded by reference scan across ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
SEED = 20240601


def rodrigues_batch(aa):
    th = np.linalg.norm(aa, axis=1)
    K = np.zeros((aa.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -aa[:, 2], aa[:, 1], aa[:, 2], -aa[:, 0], -aa[:, 1], aa[:, 0]
    th = np.maximum(th, 1e-12)
    A = (np.sin(th) / th)[:, None, None]; B = ((1 - np.cos(th)) / th ** 2)[:, None, None]
    return np.eye(3)[None] + A * K + B * (K @ K)


def trajectory(F):
    """T_lw parameter blocks of F scans: 0.1 m steps along +z, 0.5 deg yaw per step."""
    k = np.arange(F)
    yaw = np.deg2rad(0.5) * k
    aa_wl = np.stack([np.zeros(F), yaw, np.zeros(F)], axis=1)       # y is the vertical axis (x right, y down, z fwd)
    t_wl = np.stack([np.zeros(F), np.zeros(F), 0.1 * k], axis=1)
    R_wl = rodrigues_batch(aa_wl)
    aa_lw = -aa_wl
    t_lw = -np.einsum("fji,fj->fi", R_wl, t_wl)
    return aa_lw, t_lw


def pair_list(F, nb):
    """Ordered (ref, nei) pairs: each scan against its nb temporally nearest scans, both directions."""
    ref, nei = [], []
    half = nb // 2
    for i in range(F):
        cand = [i + d for d in range(-half, half + 1) if d != 0]
        cand = [c for c in cand if 0 <= c < F]
        j = 1
        while len(cand) < nb and (i - half - j >= 0 or i + half + j < F):
            for c in (i - half - j, i + half + j):
                if 0 <= c < F and len(cand) < nb:
                    cand.append(c)
            j += 1
        for c in cand:
            ref.append(i); nei.append(c)
    return np.array(ref, np.int32), np.array(nei, np.int32)


def synth_records(rng, aa, t, ref, nei, rows_per_pair):
    """ICP-like point-to-plane records for each pair (vectorised): neighbour-frame point in a
    12 x 8 x 3 m room, unit plane normal in the reference frame, signed offset ~ N(0, 2 cm)."""
    P = len(ref)
    R = rodrigues_batch(aa)
    out = np.empty((P * rows_per_pair, 7))
    for p in range(P):
        n = rows_per_pair
        Pn = rng.uniform([-6, -1.5, -4], [6, 1.5, 4], size=(n, 3))
        Rrn = R[ref[p]] @ R[nei[p]].T
        Pr = (Pn - t[nei[p]]) @ Rrn.T + t[ref[p]]
        nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        d = -(nrm * Pr).sum(1) + rng.normal(size=n) * 0.02
        o = out[p * n:(p + 1) * n]
        o[:, 0:3] = Pn; o[:, 3:6] = nrm; o[:, 6] = d
    off = np.arange(P + 1, dtype=np.int64) * rows_per_pair
    return out, off


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scans", type=int, default=64, help="number of scans F in the batch")
    ap.add_argument("--neighbors", type=int, default=8, help="ordered pairs per reference scan")
    ap.add_argument("--points", type=int, default=65536, help="points per scan (16 x 4096)")
    ap.add_argument("--functor", choices=["angle", "meter"], default="angle")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run)"
    dev = torch.device("cuda", local_rank)

    import panovlm_amd as pv
    ctx = pv.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---- workload (identical on every rank; each rank keeps its shard of reference scans) --------
    F, nb = args.scans, args.neighbors
    aa_true, t_true = trajectory(F)
    rng = np.random.default_rng(SEED)
    aa0 = aa_true + rng.uniform(-np.deg2rad(0.5), np.deg2rad(0.5), size=(F, 3))
    t0 = t_true + rng.uniform(-0.02, 0.02, size=(F, 3))
    ref_all, nei_all = pair_list(F, nb)
    lo, hi = (F * rank) // world, (F * (rank + 1)) // world
    mine = (ref_all >= lo) & (ref_all < hi)
    ref, nei = ref_all[mine], nei_all[mine]
    rng_r = np.random.default_rng(SEED + 1 + rank)
    rows, off = synth_records(rng_r, aa0, t0, ref, nei, args.points)
    kind = pv.POINT2PLANE_ANGLE if args.functor == "angle" else pv.POINT2PLANE_METER
    flags = pv.FLAG_NORMALIZE_DISTANCE
    loss_a = 2 * np.pi / 180 if args.functor == "angle" else 0.2
    rs = pv.ResidualSet.upload(ctx, kind, rows, off, ref, nei, flags=flags, weight=1.0)
    n_local = rs.n
    up = sorted({(min(a, b), max(a, b)) for a, b in zip(ref_all.tolist(), nei_all.tolist())})
    neq = pv.NormalEq(ctx, F, [u[0] for u in up], [u[1] for u in up])
    packed = torch.zeros(neq.size, dtype=torch.float64, device=dev)
    d_aa = torch.from_numpy(np.ascontiguousarray(aa0)).to(dev)
    d_t = torch.from_numpy(np.ascontiguousarray(t0)).to(dev)
    n_total = torch.tensor([n_local], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(n_total)
    n_total = int(n_total.item())

    def step(i):
        # a fresh parameter point every step (as an LM iteration does): tiny deterministic nudge
        d_t.add_(1e-7)
        ctx.set_poses_dev(F, d_aa.data_ptr(), d_t.data_ptr())
        neq.accumulate_dev(rs, packed.data_ptr(), pv.LOSS_HUBER, loss_a, zero_first=True)
        if world > 1:
            dist.all_reduce(packed)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    ctx.profile_enable(True)
    t0_wall = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0_wall
    kern_ms, kern_n = ctx.profile_read(0)
    ctx.profile_enable(False)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    # materialise mode (r + 1x12 J written to HBM, the Ceres-feeding path) — reported, not `value`
    d_r = torch.empty(max(n_local, 1), dtype=torch.float64, device=dev)
    d_J = torch.empty((max(n_local, 1), 12), dtype=torch.float64, device=dev)
    rs.eval_dev(d_r.data_ptr(), d_J.data_ptr())
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    for _ in range(5):
        rs.eval_dev(d_r.data_ptr(), d_J.data_ptr())
    mat_ms, mat_n = ctx.profile_read(1)
    ctx.profile_enable(False)
    del d_J

    if rank == 0:
        bytes_per_eval = 8 * 7  # 7 fp64 SoA columns; pair ids are per segment, not per row
        k_avg_s = kern_ms / max(kern_n, 1) * 1e-3
        achieved = n_local * bytes_per_eval / k_avg_s / 1e9
        out = {
            "metric": "M residual+Jacobian evals/sec (point-to-plane, 64k pts/scan)",
            "value": n_total * args.steps / dt / 1e6,
            "unit": "M evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "synthetic VLP-16 16x4096 scan pairs: %d scans x %d neighbours = %d ordered pairs, %d residual blocks; "
                            "Point2Plane_%s%s + HuberLoss, fused r+J -> per-pose 6x6/6x1 blocks%s" % (
                                F, nb, len(ref_all), n_total, "Angle" if args.functor == "angle" else "Meter",
                                "(normalize_distance)" if args.functor == "angle" else "",
                                ", RCCL all-reduce of %d doubles per step" % neq.size if world > 1 else ""),
                "mode": "fused-normal-equations", "functor": args.functor, "pairs": int(len(ref_all)), "rows_per_pair": args.points,
                "source": "synthetic correspondence records"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": "k_eval_fused", "kernel_avg_ms": kern_ms / max(kern_n, 1), "launches": kern_n,
                         "bytes_per_eval": bytes_per_eval, "evals_per_launch": n_local,
                         "ceiling_M_evals_per_s_at_64B": HBM_PEAK_GBPS * 1e9 / 64 / 1e6},
            "materialise": {"kernel": "k_eval_materialise", "kernel_avg_ms": mat_ms / max(mat_n, 1),
                            "M_evals_per_s": n_local / (mat_ms / max(mat_n, 1) * 1e-3) / 1e6,
                            "GBps": n_local * (bytes_per_eval + 104) / (mat_ms / max(mat_n, 1) * 1e-3) / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rows, off, ref, nei, aa0, t0, kind, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(rows, off, ref, nei, aa, t, kind, budget_s):
    """The CPU oracle (restated reference algorithm: Jet<12> AutoDiff through the reference's rotation
    chain, base/CostFunction.h) timed on this box's host cores on a bounded sample of the same rows."""
    from oracle import oracle as orc
    threads = orc.num_threads()
    n_probe = min(rows.shape[0], 200_000)
    rid = np.repeat(ref, np.diff(off))[:rows.shape[0]].astype(np.int32)
    nid = np.repeat(nei, np.diff(off))[:rows.shape[0]].astype(np.int32)
    orows = np.concatenate([rows, np.ones((rows.shape[0], 1))], axis=1)
    t0 = time.perf_counter()
    orc.evaluate(kind, orows[:n_probe], rid[:n_probe], nid[:n_probe], aa, t, normalize=True, jac=True)
    probe = time.perf_counter() - t0
    n = int(min(rows.shape[0], max(n_probe, n_probe * budget_s / max(probe, 1e-6))))
    t0 = time.perf_counter()
    orc.evaluate(kind, orows[:n], rid[:n], nid[:n], aa, t, normalize=True, jac=True)
    dt = time.perf_counter() - t0
    return {"value": n / dt / 1e6, "unit": "M evals/s", "cores": threads, "kind": "port",
            "sample": "%d residual blocks (first rows of the same batch), r + 1x12 J by Jet<12> AutoDiff, OpenMP %d threads, %.1f s" % (n, threads, dt)}


if __name__ == "__main__":
    main()
