#!/usr/bin/env python3
"""bench.py — M residual+Jacobian evals/s of the point-to-plane hot path on MI355X.

Workload (BASELINE.json configs[1]): F synthetic VLP-16 scans of 16 x 4096 = 65 536
points, every scan associated against its `--neighbors` temporally nearest scans (ordered pairs, both
directions) by the voxel-hash k-NN + plane-fit kernels (AssociatePoint2Plane, k = 10, thr 1.0 m,
tol 0.05).  The accepted correspondences stay in HBM as a fp64 SoA residual set.

One *step* = one Gauss-Newton / LM linearisation pass over the whole batch at a fresh parameter
point: refresh the pose table, evaluate every Point2Plane_Angle(normalize_distance) residual and its
1 x 12 Jacobian, apply HuberLoss(2 deg), contract into the per-pose 6x6 / 6x1 normal-equation blocks
(fused, nothing but the blocks is written) and, for N > 1, all-reduce the packed block buffer over
RCCL.  `value` = residual+Jacobian evaluations of all ranks / wall time of the K timed steps.

Scaling is STRONG: the batch is fixed and sharded by reference scan across ranks.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run
(one rank per GPU).  Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s achievable
# The one table every `roof` object of this file is priced against (/opt/skills/guides/MI355X_MICROARCH.md; the link rate is the pcie block's measurement).
PEAKS = {
    "hbm_GBps": HBM_PEAK_GBPS,
    # a SIMD issues one wave64 VALU instruction per 4 cycles — fp32, integer and v_min/max_f64 alike (measured at > 90 % busy: 4.3-4.5 cycles,
    # profiles/r5_assoc_variants.txt section 1): 256 CUs x 4 SIMDs x 2.4 GHz / 4
    "valu_wave_insts_per_s": 1024 * 2.4e9 / 4.0,
    "host_link_GBps": 55.7,
}


def roof(bound, achieved, unit, issue_peak=None, algorithmic_peak=None, **extra):
    """One shape for the side blocks' roofs.  Two fractions, kept apart on purpose:
      issue_efficiency     achieved / the rate at which the kernel's OWN instruction stream (SQ_INSTS_VALU) could issue — how well the loop as written
                           keeps the SIMDs busy; a bloated loop scores the same (what `frac` meant for these blocks up to round 5; `frac` is kept as its alias)
      frac_of_algorithmic  achieved / the rate an implementation executing only the operations the PROBLEM needs would reach at full issue (the floor is
                           stated beside it): distance from the problem, not from the loop."""
    out = {"bound": bound, "achieved": achieved, "unit": unit}
    if issue_peak:
        out["at_full_issue_of_own_instructions"] = issue_peak
        out["issue_efficiency"] = achieved / issue_peak if achieved is not None else None
        out["frac"] = out["issue_efficiency"]
    if algorithmic_peak:
        out["at_full_issue_of_algorithmic_floor"] = algorithmic_peak
        out["frac_of_algorithmic"] = achieved / algorithmic_peak if achieved is not None else None
    out.update(extra)
    return out


def cpu_quota():
    """CPUs this process may use on average: the cgroup's CFS quota (cpu.max: "<quota us> <period us>") when there is one, else None.  The GPU boxes of this
    pool show 256 hardware threads and a quota of 16: more threads than that only run in bursts until the period's budget is spent, and then stall."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                return float(quota) / float(period)
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / p
    except Exception:
        pass
    return None


def usable_cpus():
    """Hardware threads, or the cgroup's CPU quota when that is smaller."""
    n = os.cpu_count() or 8
    q = cpu_quota()
    return max(1, min(n, int(q + 0.5))) if q else n


def _gen(args):
    from panovlm_amd import synthetic as sy
    k, cols, voxel = args
    s = sy.make_scan(k, cols=cols, downsample_targets=voxel)
    s.pop("local_xyz")
    return s


def generate_scans(ids, cols, voxel):
    import multiprocessing as mp
    ids = list(ids)
    procs = max(1, min(len(ids), usable_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))), 64))
    if procs == 1:
        return {k: _gen((k, cols, voxel)) for k in ids}
    with mp.get_context("fork").Pool(procs) as pool:
        out = pool.map(_gen, [(k, cols, voxel) for k in ids], chunksize=1)
    return dict(zip(ids, out))


class Watchdog:
    """A collective that never returns would cost the only multi-GPU record this project can get: every phase of a multi-rank
    run that can block on another rank (rendezvous, first collectives, the timed region) runs under a deadline.  When it expires
    the process says which phase hung and exits with code 3 (os._exit: a rank stuck inside a collective cannot unwind)."""

    def __init__(self, rank):
        self.rank, self.timer, self.phase = rank, None, ""

    def arm(self, phase, seconds):
        import threading
        self.disarm()
        self.phase = phase

        def fire():
            sys.stderr.write("[bench] rank %d: WATCHDOG — '%s' did not finish within %d s; aborting the run\n" % (self.rank, phase, seconds))
            sys.stderr.flush()
            if self.rank == 0:
                print(json.dumps({"metric": "M residual+Jacobian evals/sec (point-to-plane, 64k pts/scan)", "value": None, "unit": "M evals/s",
                                  "error": "watchdog: '%s' hung for %d s" % (phase, seconds)}), flush=True)
            os._exit(3)
        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def visible_gpus():
    """GPUs this process may use, counted without initialising HIP in it (the ranks are still to be forked): rocm-smi-free, through
    a child interpreter."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def launch_ranks(n, shared_gpu):
    """Re-executes this script as n ranks under torch.distributed.run on 127.0.0.1 (free port) and returns its exit code."""
    import socket
    import subprocess
    have = visible_gpus()
    if not shared_gpu and have < n:
        sys.stderr.write("bench.py: --gpus %d needs %d GPUs, this node shows %d (PVLM_BENCH_SHARED_GPU=1 runs the %d ranks on one GPU as a functional check)\n"
                         % (n, n, have, n))
        return 2
    if shared_gpu and have < 1:
        sys.stderr.write("bench.py: no GPU visible\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // n)))
    env.setdefault("PVLM_HOST_THREADS", str(max(2, min(16, usable_cpus() // n))))     # staging / table passes of libpvlm.so: the ranks share the node's CPU budget
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scans", type=int, default=2048, help="number of scans F in the batch")
    ap.add_argument("--neighbors", type=int, default=8, help="ordered pairs per reference scan")
    ap.add_argument("--cols", type=int, default=4096, help="azimuth steps per ring (16 rings)")
    ap.add_argument("--functor", choices=["angle", "meter"], default="angle")
    ap.add_argument("--targets", choices=["voxel", "raw"], default="voxel",
                    help="surfLessFlat target cloud: 0.2 m voxel-grid centroids of the scan (what the reference's extractor "
                         "produces) or the raw 65 536 points (BASELINE.md §2 wording; 94%% of the queries are then rejected by "
                         "the reference's own collinearity test because their 10-NN lie on one ring)")
    ap.add_argument("--tolerance", type=float, default=0.05, help="lidar_plane_tolerance (Room 0.05, Floor 0.01)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5, help="untimed steady-state pre-warm before the W warm-up steps (clock ramp)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step as a captured HIP graph (pvlm_graph_*).  auto = off: measured on MI355X / ROCm 7.0 a replayed graph "
                         "of the five kernels costs 28-33 us around the fused kernel against 22-26 us for five plain launches "
                         "(per_rank_projection reports both), and eager launches let every launch of the dominant kernel inside the "
                         "timed region be bracketed by HIP events (roofline.achieved)")
    ap.add_argument("--no-mvs", action="store_true", help="skip the panoramic MVS block (resident views at 1440 x 720 and 5760 x 2880)")
    ap.add_argument("--no-projection", action="store_true", help="skip the per-rank projection block (scans/2, /4, /8 on this GPU)")
    ap.add_argument("--collective-timeout", type=float, default=float(os.environ.get("PVLM_BENCH_COLLECTIVE_TIMEOUT", "240")),
                    help="N > 1: seconds any phase that waits for other ranks may take before the watchdog aborts the run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # functional check of the multi-rank control flow on a ONE-GPU box (not a measurement): all ranks share GPU 0 and the
    # exchange goes through gloo — PVLM_BENCH_SHARED_GPU=1 python bench.py --gpus 2
    shared_gpu = os.environ.get("PVLM_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher of its own N ranks (one per GPU, RCCL) —
        # the same command line torch.distributed.run would be given by hand
        sys.exit(launch_ranks(args.gpus, shared_gpu))
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d, or run `python bench.py --gpus %d` and let it spawn its ranks\n"
                         % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)

    from panovlm_amd import sharding
    from panovlm_amd import synthetic as sy
    F, nb = args.scans, args.neighbors
    ref_all, nei_all = sy.pair_list(F, nb)
    ref, nei = sharding.shard_pairs(ref_all, nei_all, F, rank, world)   # block partition by reference scan
    needed = sorted(set(ref.tolist()) | set(nei.tolist()))
    t_gen = time.perf_counter()
    scans = generate_scans(needed, args.cols, 0.2 if args.targets == "voxel" else 0.0)       # before torch/HIP initialise (fork-safe)
    t_gen = time.perf_counter() - t_gen

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dog = Watchdog(rank)
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC (RCCL across processes)
        dog.arm("rendezvous (init_process_group)", args.collective_timeout)
        to = datetime.timedelta(seconds=args.collective_timeout)
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=to)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=to)
        dog.disarm()
    dev = torch.device("cuda", local_rank)

    import panovlm_amd as pv
    ctx = pv.Context(local_rank)
    # one real (capturable) stream carries everything: the library's kernels, torch's few element-wise ops and the all-reduce
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    ctx.set_stream(side.cuda_stream)

    kind = pv.POINT2PLANE_ANGLE if args.functor == "angle" else pv.POINT2PLANE_METER
    loss_a = 2 * np.pi / 180 if args.functor == "angle" else 0.2     # util/Optimization.cpp:515-517
    n_queries = int(sum(scans[int(n)]["flat_xyz"].shape[0] for n in nei))
    n_targets = int(sum(scans[int(r)]["less_xyz"].shape[0] for r in ref))
    # Working set reserved up front, like the scans are uploaded up front: hipMalloc maps pages at 40-70 ms per GB
    # (tools/micro/alloc_cost.hip), the association's output alone is 56 B x accepted rows.  Upper bound: every query
    # accepted + the two pipeline slots of scratch + the scans.
    t_res = time.perf_counter()
    shards_extra = 0.9 if (world == 1 and not args.no_projection and args.scans >= 64) else 0.0     # rank-0 shards of N = 2, 4, 8 live beside the batch
    reserve_bytes = int(n_queries * 56 * (1.02 + shards_extra)) + 2 * min(n_queries, 16 << 20) * 100 + sum(
        (len(s["flat_xyz"]) * 16 + len(s["less_xyz"]) * 48 + (4 << 20)) for s in scans.values()) + (1 << 30)
    info = ctx.device_info()
    reserve_bytes = min(reserve_bytes, int(info["hbm_bytes"] * 0.8))
    ctx.reserve(reserve_bytes)
    t_res = time.perf_counter() - t_res
    # one pvlm_scan_upload_batch for the rank's scans (one slab, one set of grid-build launches), in slices that keep the staged front under 2 GB
    dscans, keys_, per = {}, sorted(scans), max(1, (2 << 30) // (16 * args.cols * 40))
    for a0 in range(0, len(keys_), per):
        part = keys_[a0:a0 + per]
        dscans.update(zip(part, pv.Scan.upload_batch(ctx, [scans[k] for k in part])))
    ctx.synchronize()

    def associate(rr, nn_, tol, ds=dscans):
        return ctx.assoc_point2plane([ds[int(r)] for r in rr], [ds[int(n)] for n in nn_], tol, 1.0, kind=kind, flags=pv.FLAG_NORMALIZE_DISTANCE, weight=1.0)

    def timed_assoc(rr, nn_, tol, ds=dscans):
        ctx.synchronize()
        m0 = ctx.mem_info()
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        r_ = associate(rr, nn_, tol, ds)
        ctx.synchronize()
        wall = time.perf_counter() - t0
        ms, launches = ctx.profile_read(2)
        ctx.profile_enable(False)
        return r_, wall, ms, launches, ctx.mem_info()["device_allocs"] - m0["device_allocs"]

    rs, t_assoc_first, assoc_ms_first, _, allocs_first = timed_assoc(ref, nei, args.tolerance)
    # the reference re-associates at every outer iteration (lidar_mapping/LidarOdometry.cpp:38-110): the steady state
    # of the call is the SECOND and later ones, with the previous residual set given back first
    rs.close()
    rs, t_assoc, assoc_ms, assoc_n, allocs_steady = timed_assoc(ref, nei, args.tolerance)
    n_local = rs.n

    poses = [sy.pose_params(*sy.estimated_pose(k)) for k in range(F)]
    aa0 = np.array([p[0] for p in poses]); t0 = np.array([p[1] for p in poses])
    ui, uj = sharding.unordered_pairs(ref_all, nei_all)                   # same block structure on every rank
    neq = pv.NormalEq(ctx, F, ui, uj)
    packed = torch.zeros(neq.size, dtype=torch.float64, device=dev)
    d_aa = torch.from_numpy(np.ascontiguousarray(aa0)).to(dev)
    d_t = torch.from_numpy(np.ascontiguousarray(t0)).to(dev)
    tot = torch.tensor([n_local, n_queries], dtype=torch.int64, device=dev)
    if world > 1:
        # the first collective of the run: set-up times differ between ranks (scan generation, association), so the deadline is generous
        dog.arm("first all-reduce (block counts)", max(args.collective_timeout, 600.0))
        dist.all_reduce(tot)
        torch.cuda.synchronize()
        dog.disarm()
    n_total, q_total = int(tot[0].item()), int(tot[1].item())

    # ---- the step: k_pose_table -> k_pair_table -> k_eval_fused -> k_pair_epilogue -> k_neq_gather (-> all-reduce) ----
    # Captured once as a HIP graph (pvlm_graph_*) and replayed: one submission per LM step instead of five launches
    # (+ the collective) — optional (--graph on): a replayed graph measured SLOWER than the eager launches.
    # Default: torch.distributed's communicator (backend "nccl" = RCCL), the path every multi-GPU PyTorch job exercises.
    # PVLM_BENCH_COMM=pvlm routes the all-reduce through the library's own communicator (pvlm_comm_*, what the C++ hosts
    # use) on the context stream instead; it is opt-in because no multi-GPU box was available to this repository's
    # author to run it on (the 2-rank tests share one GPU, which RCCL refuses), and a collective that misbehaves hangs.
    comm, comm_mode = None, "none"
    if world > 1:
        comm_mode = "torch"
        if not shared_gpu and os.environ.get("PVLM_BENCH_COMM", "torch") == "pvlm":
            # every rank must take the same branch: the id travels even when rank 0 failed to make one, and the ranks agree on
            # the outcome of the creation before anybody uses the communicator
            ids = [None]
            if rank == 0:
                try:
                    ids = [ctx.comm_unique_id()]
                except Exception as e:
                    sys.stderr.write("[bench] pvlm_comm_unique_id failed (%s)\n" % str(e)[:200])
            dist.broadcast_object_list(ids, src=0)
            made = 0
            if ids[0] is not None:
                try:
                    comm = pv.Comm(ctx, world, rank, ids[0]); made = 1
                except Exception as e:
                    sys.stderr.write("[bench] pvlm_comm_create failed on rank %d (%s)\n" % (rank, str(e)[:200]))
            flag = torch.tensor([made], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                comm_mode = "pvlm"
            else:
                if comm is not None:
                    comm.close()
                comm = None
                sys.stderr.write("[bench] pvlm_comm unavailable on some rank; using torch.distributed\n")

    use_graph = args.graph == "on"

    def make_step(rs_, neq_, packed_, with_comm, graphed=None):
        graphed = use_graph if graphed is None else graphed

        def local():
            ctx.set_poses_dev(F, d_aa.data_ptr(), d_t.data_ptr())
            neq_.accumulate_dev(rs_, packed_.data_ptr(), pv.LOSS_HUBER, loss_a, zero_first=True)
            if with_comm and comm is not None:
                comm.allreduce_sum_f64(packed_.data_ptr(), neq_.size)
        graph = None
        if graphed:
            try:
                local(); ctx.synchronize()          # first run binds / sizes everything: nothing left to allocate inside the capture
                ctx.graph_begin()
                try:
                    local()
                finally:
                    graph = ctx.graph_end()
            except Exception as e:
                sys.stderr.write("[bench] graph capture failed (%s); eager launches\n" % str(e)[:200])
                graph = None

        def step():
            d_t.add_(1e-9)   # a fresh parameter point every step, as an LM iteration has
            if graph is not None:
                graph.launch()
            else:
                local()
            if with_comm and world > 1 and comm is None:
                dist.all_reduce(packed_)
        return step, graph

    step, graph = make_step(rs, neq, packed, True)
    if world > 1 and comm is not None:
        # the captured collective must give what torch's eager one gives
        step(); torch.cuda.synchronize()
        a = packed.clone()
        ctx.set_poses_dev(F, d_aa.data_ptr(), d_t.data_ptr())
        neq.accumulate_dev(rs, packed.data_ptr(), pv.LOSS_HUBER, loss_a, zero_first=True)
        dist.all_reduce(packed); torch.cuda.synchronize()
        ok = torch.tensor([1 if torch.allclose(a, packed, rtol=1e-9, atol=1e-9 * float(packed.abs().max())) else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            sys.stderr.write("[bench] pvlm_comm all-reduce disagrees with torch.distributed; using torch\n")
            comm = None; comm_mode = "torch(fallback)"
            step, graph = make_step(rs, neq, packed, True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clocks: MI355X ramps memory/fabric clocks under sustained load; a 1 ms step measured after three warm-up steps runs
    # ~10 % below its steady state (measured: 89.7 -> 99.4 G eval/s at 101 M evals/launch).  Untimed pre-warm, then the
    # W warm-up steps of the contract.  The pre-warm uses the same step (collective included: the loop count is fixed).
    local_step, _ = (step, None) if world == 1 else make_step(rs, neq, torch.zeros_like(packed), False)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_seconds:   # local work only: the loop count differs between ranks
        for _ in range(8):
            local_step()
        torch.cuda.synchronize()
    if world > 1:
        dog.arm("warm-up + timed steps (all-reduce of %d doubles per step)" % neq.size, max(args.collective_timeout, 60.0 + 2.0 * (args.steps + args.warmup)))
    for _ in range(max(args.warmup, 1) if world > 1 else args.warmup):
        step()
    fence()
    # eager steps (N = 1): every launch of the fused kernel inside the timed region is bracketed by HIP events on the
    # stream it runs on.  Graph replay (N > 1): events cannot be recorded inside a replayed graph, so the fused kernel is
    # timed by an eager pass of the same K launches right after the timed region (same stream, data and clocks).
    if graph is None:
        ctx.profile_enable(True)
    t_wall = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t_wall
    if graph is not None:
        ctx.profile_enable(True)
        for _ in range(args.steps):
            d_t.add_(1e-9)
            ctx.set_poses_dev(F, d_aa.data_ptr(), d_t.data_ptr())
            neq.accumulate_dev(rs, packed.data_ptr(), pv.LOSS_HUBER, loss_a, zero_first=True)
    kern_ms, kern_n = ctx.profile_read(0)
    ctx.profile_enable(False)
    if graph is not None:
        step(); torch.cuda.synchronize()          # leave the reduced buffer of a full step behind
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    # the summed system at the INITIAL parameter point (the timed steps move the point by 1e-9 each and the pre-warm runs a clock-dependent number of
    # them), as two numbers a one-rank run and an N-rank run of the same command must agree on (to the order of the sums)
    d_t.copy_(torch.from_numpy(np.ascontiguousarray(t0)).to(dev))
    ctx.set_poses_dev(F, d_aa.data_ptr(), d_t.data_ptr())
    neq.accumulate_dev(rs, packed.data_ptr(), pv.LOSS_HUBER, loss_a, zero_first=True)
    if world > 1:
        if comm is not None:
            comm.allreduce_sum_f64(packed.data_ptr(), neq.size)
        else:
            dist.all_reduce(packed)
    torch.cuda.synchronize()
    packed_sum, packed_l1 = float(packed.sum().item()), float(packed.abs().sum().item())
    cost = float(packed[-1].item())
    dog.disarm()
    # N > 1: the all-reduce of the packed blocks alone (barrier-bracketed, max over ranks) and the per-rank load table
    allreduce_us, rank_table = None, None
    if world > 1:
        dog.arm("all-reduce timing", args.collective_timeout)
        scratch = torch.zeros_like(packed)
        reps = 50

        def one():
            if comm is not None:
                comm.allreduce_sum_f64(scratch.data_ptr(), neq.size)
            else:
                dist.all_reduce(scratch)
        for _ in range(5):
            one()
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            one()
        fence()
        ar = torch.tensor([(time.perf_counter() - t0) / reps * 1e6], dtype=torch.float64, device=dev)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        allreduce_us = float(ar.item())
        table = torch.zeros((world, 4), dtype=torch.float64, device=dev)
        table[rank, 0] = len(ref); table[rank, 1] = n_queries; table[rank, 2] = n_local; table[rank, 3] = kern_ms / max(kern_n, 1)
        dist.all_reduce(table)
        rank_table = [{"rank": r, "pairs": int(v[0]), "queries": int(v[1]), "evals": int(v[2]), "fused_kernel_ms": v[3]} for r, v in enumerate(table.tolist())]
        dog.disarm()

    # ---- what one rank of an N-GPU run has to do, measured on this one GPU (no 8-GPU node is mine to launch) ----
    projection = None
    if rank == 0 and world == 1 and not args.no_projection and args.scans >= 64:
        try:
            projection = per_rank_projection(ctx, pv, torch, sharding, args, associate, make_step, ref_all, nei_all, F, ui, uj, dev,
                                             dt / args.steps * 1e3, kern_ms / max(kern_n, 1), neq.size)
        except Exception as e:  # reporting extra only
            projection = {"error": str(e)[:200]}

    extra_assoc = None
    if rank == 0 and world == 1:
        try:
            extra_assoc = association_points(ctx, pv, torch, args, scans, dscans, ref, nei, associate, timed_assoc, make_step, F, ui, uj, dev)
        except Exception as e:  # reporting extra only
            extra_assoc = {"error": str(e)[:200]}

    # materialise mode (r + 1x12 J written to HBM: the Ceres-feeding path) — reported, never `value`
    mat = None
    if rank == 0:
        try:
            d_r = torch.empty(max(n_local, 1), dtype=torch.float64, device=dev)
            d_J = torch.empty((max(n_local, 1), 12), dtype=torch.float64, device=dev)
            rs.eval_dev(d_r.data_ptr(), d_J.data_ptr())
            torch.cuda.synchronize()
            ctx.profile_enable(True)
            for _ in range(5):
                rs.eval_dev(d_r.data_ptr(), d_J.data_ptr())
            mat_ms, mat_n = ctx.profile_read(1)
            ctx.profile_enable(False)
            mat = {"kernel": "k_eval_materialise", "kernel_avg_ms": mat_ms / max(mat_n, 1),
                   "M_evals_per_s": n_local / (mat_ms / max(mat_n, 1) * 1e-3) / 1e6,
                   "GBps": n_local * (56 + 104) / (mat_ms / max(mat_n, 1) * 1e-3) / 1e9, "bytes_per_eval": 160}
            del d_J, d_r
        except Exception as e:  # reporting extra only
            mat = {"error": str(e)[:200]}

    # the Ceres-feeding boundary: r + J made HOST-visible (pinned memory, asynchronous copies) — PCIe-bound, reported, never `value`
    pcie = None
    if rank == 0 and world == 1:
        try:
            pcie = pcie_block(ctx, pv, associate, ref, nei, args)
        except Exception as e:  # reporting extra only
            pcie = {"error": str(e)[:200]}

    # 5.7K equirectangular panorama (5760 x 2880): whole-image CamToImage / ImageToCam maps (K7) and the
    # camera<->LiDAR line-association voting loop (K8) for a Room-sized batch — reported beside the headline, never `value`
    pano = None
    if rank == 0 and world == 1:
        try:
            pano = panorama_block(ctx, pv, torch, dev)
        except Exception as e:  # reporting extra only
            pano = {"error": str(e)[:200]}
    # N > 1: the image-space paths shard per view with no exchange at all (sharding.shard_views): every rank maps its own
    # 5.7K panorama (K7) and runs the PatchMatch kernels on its own 5.7K view (K11, one K13 iteration); the job's rate is
    # the sum of the ranks' rates.  Every rank takes part in the one all-reduce whatever happened to its measurement.
    image_space = None
    if world > 1:
        dog.arm("image-space block (per-view shards, one all-reduce of the rates)", max(args.collective_timeout, 600.0))
        local = [0.0, 0.0, 0.0, 0.0, 0.0]
        try:
            pb = panorama_block(ctx, pv, torch, dev, with_votes=False)
            local[0] = pb["cam_to_image_f32"]["G_points_per_s"]; local[1] = pb["image_to_cam_f32"]["G_points_per_s"]
            if not args.no_mvs:
                mv = mvs_one_size(ctx, 2880, 5760, k13_reps=1)
                local[2] = mv["k11_scoring_pass"]["M_pixels_per_s"]; local[3] = mv["k13_patchmatch_iteration"]["M_pixels_per_s"]
            local[4] = 1.0
        except Exception as e:  # reporting extra only
            sys.stderr.write("[bench] rank %d: image-space block failed (%s)\n" % (rank, str(e)[:200]))
        tsum = torch.tensor(local, dtype=torch.float64, device=dev)
        dist.all_reduce(tsum)
        dog.disarm()
        if rank == 0:
            v = tsum.tolist()
            image_space = {"sharding": "one 5.7K view per rank, no exchange (weak scaling by construction)", "ranks_measured": int(round(v[4])),
                           "cam_to_image_f32_G_points_per_s": v[0], "image_to_cam_f32_G_points_per_s": v[1],
                           "cam_to_image_f32_frac_of_hbm_peak_per_gpu": v[0] * 20 / HBM_PEAK_GBPS / max(v[4], 1.0),
                           "image_to_cam_f32_frac_of_hbm_peak_per_gpu": v[1] * 20 / HBM_PEAK_GBPS / max(v[4], 1.0),
                           "mvs_k11_M_pixels_per_s": v[2], "mvs_k13_iteration_M_pixels_per_s": v[3], "rank0": {"cam_to_image_f32_G_points_per_s": local[0], "image_to_cam_f32_G_points_per_s": local[1]}}

    # the stage in front of the path: ReOrderVLP + Segmentation + adaptive curvature of a Room-sized batch of raw scans
    features = None
    if rank == 0 and world == 1 and not args.no_mvs:
        try:
            features = features_block(ctx, pv)
        except Exception as e:  # reporting extra only
            features = {"error": str(e)[:200]}
        try:
            features["undistort"] = undistort_block(ctx, pv)
        except Exception as e:
            features["undistort"] = {"error": str(e)[:200]}
        try:
            features["extract_features_batch"] = extract_features_batch_block()
        except Exception as e:
            features["extract_features_batch"] = {"error": str(e)[:200]}
    pose_solve = None
    if rank == 0 and world == 1 and not args.no_mvs:
        try:
            pose_solve = pose_solve_block(ctx)
        except Exception as e:  # reporting extra only
            pose_solve = {"error": str(e)[:200]}
    mvs = None
    if rank == 0 and world == 1 and not args.no_mvs:
        try:
            mvs = mvs_block(ctx, pv)
        except Exception as e:  # reporting extra only
            mvs = {"error": str(e)[:200]}

    if rank == 0:
        # HBM traffic of the fused kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of
        # this same command, summarised by tools/pmc_traffic.py into profiles/): used only when it was collected on
        # exactly this workload (the synthetic batch is deterministic, so evals_per_launch identifies it).
        traffic, traffic_src, traffic_stamp = None, None, None
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json"))):
            try:
                pm = json.load(open(f))
                if pm.get("evals_per_launch") == n_local and world == 1:
                    for k, v in pm["kernels"].items():
                        if "k_eval_fused<" in k:       # the headline's kernel (the wave-per-segment form runs on the short-segment side workloads)
                            traffic, traffic_src = v["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
                            traffic_stamp = {"collected_at_commit": pm.get("commit"), "collected_on": pm.get("date"),
                                             "note": "counter passes of THIS command, collected in a separate rocprofv3 --pmc run (counters cannot be read inside "
                                                     "a timed run); it is that run's traffic, cited here because the synthetic batch is deterministic"}
            except Exception:
                pass
        bytes_per_eval = 8 * 7  # 7 fp64 SoA columns (P_n, plane); pair ids live in the per-segment table
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
                "workload": "synthetic VLP-16 16x%d scans (BASELINE.md §2): %d scans x %d neighbours = %d ordered pairs, "
                            "every point a surfFlat query (%d queries), surfLessFlat targets = %s -> %d point-to-plane residual blocks "
                            "(k=10, thr=1.0 m, tol=%.2f); Point2Plane_%s + HuberLoss, "
                            "fused r+J -> per-pose 6x6/6x1 blocks%s" % (
                                args.cols, F, nb, len(ref_all), q_total,
                                "0.2 m voxel-grid centroids" if args.targets == "voxel" else "all 65 536 points", n_total, args.tolerance,
                                "Angle(normalize_distance)" if args.functor == "angle" else "Meter",
                                ", RCCL all-reduce of %d doubles per step" % neq.size if world > 1 else ""),
                "mode": "fused-normal-equations", "functor": args.functor, "targets": args.targets, "scans": F, "pairs": int(len(ref_all)),
                "points_per_scan": 16 * args.cols, "residual_blocks": n_total, "robust_cost": cost},
            "roofline": {"bound": "hbm", "frac": achieved / HBM_PEAK_GBPS,
                         # the same kernel on SURVEY.md §8(d)'s LITERAL workload (every one of the 65 536 points a target: 94 % of the queries are
                         # rejected by the reference's collinearity test, the surviving segments are short) — measured below in this same run
                         "frac_literal_workload": (((extra_assoc or {}).get("raw_targets") or {}).get("fused") or {}).get("frac_of_hbm_peak"),
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src, "traffic_provenance": traffic_stamp,
                         "algorithmic_bytes_per_launch": n_local * bytes_per_eval,
                         "kernel": "k_eval_fused", "kernel_avg_ms": kern_ms / max(kern_n, 1), "launches": kern_n,
                         "bytes_per_eval": bytes_per_eval, "evals_per_launch": n_local,
                         "M_evals_per_s_kernel": n_local / k_avg_s / 1e6,
                         "frac_of_64B_ceiling": (n_local / k_avg_s) / (HBM_PEAK_GBPS * 1e9 / 64),
                         "ceiling_M_evals_per_s_64B": HBM_PEAK_GBPS * 1e9 / 64 / 1e6,
                         "ceiling_M_evals_per_s_56B": HBM_PEAK_GBPS * 1e9 / 56 / 1e6,
                         "literal_workload": "association.raw_targets.fused: %s" % ((((extra_assoc or {}).get("raw_targets") or {}).get("fused") or {}).get("what"))},
            "association": association_block(n_queries, n_targets, n_local, int(len(ref)), assoc_ms, assoc_n, t_assoc, t_assoc_first, assoc_ms_first,
                                             allocs_first, allocs_steady, t_res, reserve_bytes, ctx.mem_info(), extra_assoc),
            "step": {"graph": graph is not None, "comm": comm_mode, "packed_sum": packed_sum, "packed_l1": packed_l1, "allreduce_doubles": int(neq.size) if world > 1 else 0,
                     "comm_backend": None if world == 1 else ("gloo (PVLM_BENCH_SHARED_GPU: all ranks on one GPU, functional check, not a measurement)" if shared_gpu
                                                             else ("RCCL through torch.distributed (backend nccl)" if comm is None else "RCCL through pvlm_comm_*")),
                     "rccl_ranks": 0 if (world == 1 or shared_gpu) else world, "allreduce_us": allreduce_us,
                     "allreduce_us_note": None if world == 1 else "all-reduce of the packed blocks alone, %d doubles, 50 repetitions, barrier-bracketed, max over ranks" % neq.size,
                     "per_rank": rank_table, "collective_timeout_s": args.collective_timeout if world > 1 else None,
                     "kernels_per_step": 5, "ms_per_step_minus_fused_kernel": dt / args.steps * 1e3 - kern_ms / max(kern_n, 1)},
            "per_rank_projection": projection,
            "materialise": mat,
            "pcie": pcie,
            "panorama": pano,
            "mvs": mvs,
            "features": features,
            "pose_solve": pose_solve,
            "image_space_all_ranks": image_space,
            "setup": {"scan_generation_s": t_gen, "scans_generated": len(needed)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ctx, pv, dscans, ref, nei, aa0, t0, kind, args)
        # RCCL prints a version banner through C stdio; on a pipe that buffer is flushed at exit, i.e. AFTER Python's own
        # output — push it out first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dog.arm("final barrier", args.collective_timeout)
        dist.barrier()
        dog.disarm()
        dist.destroy_process_group()


def association_block(n_queries, n_targets, accepted, pairs, kernel_ms, launches, wall, wall_first, kernel_ms_first, allocs_first, allocs_steady,
                      reserve_s, reserve_bytes, mem, extra):
    """AssociatePoint2Plane for the whole batch (K2 k_knn_pairs + K3 k_fit_pairs + ordered compaction).  wall_s is the
    steady-state call (second association of the same pairs, the first result given back first — what every outer
    iteration after the first does); the first call and the up-front reservation of the working set are reported beside it."""
    comp = (n_queries + n_targets) * 16            # (Nq + Nt) x 16 B per pair, the bytes any k-NN must touch once
    out = {"kernel": "k_knn_pairs + k_fit_pairs", "pairs": pairs, "queries": n_queries, "accepted": accepted,
           "kernel_ms": kernel_ms, "launches": launches, "wall_s": wall, "wall_over_kernel": wall / max(kernel_ms * 1e-3, 1e-12),
           "M_queries_per_s_kernel": n_queries / max(kernel_ms, 1e-9) / 1e3, "M_queries_per_s_wall": n_queries / max(wall, 1e-12) / 1e6,
           "hipMalloc_calls_in_call": allocs_steady,
           "first_call": {"wall_s": wall_first, "kernel_ms": kernel_ms_first, "hipMalloc_calls": allocs_first},
           "reserve": {"seconds": reserve_s, "bytes": reserve_bytes, "pool_reserved_bytes": mem["reserved"], "pool_peak_bytes": mem["peak"]},
           "compulsory_bytes": comp, "compulsory_GBps_at_kernel_time": comp / max(kernel_ms, 1e-9) / 1e6,
           "output_bytes": accepted * 56, "scratch_bytes_per_query": 97}
    # counter evidence for K2 / K3 (separate rocprofv3 --pmc passes of this command, summarised by tools/pmc_assoc.py)
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_assoc_voxel*.json"))):      # the latest round's summary wins
        try:
            pm = json.load(open(f))
            # per-query figures: collected on the same generator at --scans 256 (134 M queries); they do not depend on the batch size
            out["pmc"] = {"source": os.path.relpath(f, ROOT), "collected_on_queries": pm.get("queries"),
                          "kernels": {k: {n: v for n, v in e.items() if n not in ("per_call", "dispatches_per_call")} for k, e in pm["kernels"].items()}}
        except Exception:
            pass
    out["kernel"] = "k_knn_pairs + k_fit_pairs (the ordered compaction runs inside k_fit_pairs since round 5; round 4's kernel figure left k_compact, 282 us per 16.7 M queries, out)"
    out["roof"] = association_roof(out.get("pmc"), out["M_queries_per_s_kernel"], "voxel")
    if extra:
        out.update(extra)
        if isinstance(out.get("raw_targets"), dict):
            raw_pmc = None
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_assoc_raw*.json"))):
                try:
                    pm = json.load(open(f)); raw_pmc = {"source": os.path.relpath(f, ROOT), "kernels": pm["kernels"]}
                except Exception:
                    pass
            out["raw_targets"]["roof"] = association_roof(raw_pmc, out["raw_targets"].get("M_queries_per_s_kernel"), "raw")
    return out


def association_roof(pmc, achieved_M_queries, which):
    """Roof of the association kernels: VALU issue (PEAKS).  issue_efficiency prices the kernels' own instruction streams (SQ_INSTS_VALU per 64 queries,
    a separate --pmc pass of tools/assoc_workload.py summarised by tools/pmc_assoc.py); frac_of_algorithmic prices a floor of the problem itself:
      K2  every candidate the pruned search has to look at (host lockstep statistics of the same search, tools/assoc_lockstep.py: candidates_per_query) costs
          3 subtractions + 3 multiplications + 2 additions (flann::L2_Simple, unfused) + 1 comparison against the 10th key = 9 operations; the ten that
          end up in the list cost at least an insertion each (19 operations: the min / max network);
      K3  per accepted-or-tested query: ten points to the reference scan's frame (10 x 3 x (3 mul + 2 add + 1 sub), unfused like upstream) = 180; the
          collinearity decision = centroid + scatter matrix (10 x 15) + a closed-form 3x3 eigen screen (~100) = 250; the plane = normal equations
          (10 x 9 fused multiply-adds) + a 3x3 solve (~60) + ten point-plane distances (10 x 4) = 190; the query's own point: 18.
    `lane_efficiency` = the share of K2's insertion-network executions a lane needs for itself."""
    issue = PEAKS["valu_wave_insts_per_s"]
    own = None
    extra = {"wave_insts_per_s_peak": issue, "cycles_per_wave_instruction": 4}
    try:
        ks = pmc["kernels"]
        per_wave = sum(v["valu_insts_per_query"] for k, v in ks.items() if k.startswith(("k_knn", "k_fit", "k_compact")))
        extra["valu_wave_insts_per_64_queries"] = {k: v["valu_insts_per_query"] for k, v in ks.items()}
        extra["simd_busy_lower_bound"] = {k: v.get("simd_valu_util_lower_bound") for k, v in ks.items()}
        own = issue * 64 / per_wave / 1e6
        extra["counters_from"] = pmc.get("source")
    except Exception:
        pass
    floor = None
    try:
        ls = json.load(open(os.path.join(ROOT, "profiles", "r5_assoc_lockstep.json")))[which]
        extra["lane_efficiency"] = ls["lane_efficiency"]
        extra["candidates_per_query"] = ls["candidates_per_query"]; extra["network_executions_per_query"] = ls["network_executions_per_query"]
        k2 = ls["candidates_per_query"] * 9 + 10 * 19
        k3 = 180 + 250 + 190 + 18
        extra["algorithmic_floor_valu_per_query"] = {"k_knn_pairs": k2, "k_fit_pairs": k3,
                                                     "note": "K3's floor is per query that reaches the fits; with raw targets 94 % of the queries stop at the collinearity decision (180 + 250)"}
        floor = issue * 64 / (k2 + k3) / 1e6
    except Exception:
        extra["lane_efficiency"] = None
    return roof("VALU issue", achieved_M_queries, "M queries/s", own, floor, **extra)


def per_rank_projection(ctx, pv, torch, sharding, args, associate, make_step, ref_all, nei_all, F, ui, uj, dev, ms_full, k6_full_ms, neq_size):
    """What rank 0 of an N-GPU run of THIS batch has to do per step, measured on this one GPU: its shard of the pair list
    (block partition by reference scan), associated and evaluated exactly like the headline, the step replayed as a
    HIP graph (as bench.py does for N > 1).  The all-reduce cannot be measured on one GPU: its size is reported, and the
    projection is given without it and with a stated assumption for it."""
    # [assumed, not measured: this block only runs at N = 1, where there is nothing to all-reduce with; a run with N > 1 reports the measured
    # figure as extra.allreduce_us] RCCL all-reduce of ~3 MB over xGMI
    assumed_allreduce_us = {2: 40.0, 4: 60.0, 8: 80.0}
    out = {"full_batch_ms_per_step": ms_full, "full_batch_fused_kernel_ms": k6_full_ms, "allreduce_bytes": int(neq_size) * 8,
           "assumed_allreduce_us": assumed_allreduce_us, "ranks": {}}
    for N in (2, 4, 8):
        r, n = sharding.shard_pairs(ref_all, nei_all, F, 0, N)
        rs_k = associate(r, n, args.tolerance)
        neq_k = pv.NormalEq(ctx, F, ui, uj)
        packed_k = torch.zeros(neq_size, dtype=torch.float64, device=dev)
        res = {"pairs": int(len(r)), "evals": int(rs_k.n)}
        for tag, graphed in (("graph", True), ("eager", False)):
            step_k, g = make_step(rs_k, neq_k, packed_k, False, graphed=graphed)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.25:          # clock ramp at short steps
                for _ in range(16):
                    step_k()
                torch.cuda.synchronize()
            reps = 100
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                step_k()
            torch.cuda.synchronize()
            res["ms_per_step_" + tag] = (time.perf_counter() - t0) / reps * 1e3
            if g is not None:
                g.close()
            elif graphed:
                res["graph_error"] = True
        ctx.profile_enable(True)
        step_k, _ = make_step(rs_k, neq_k, packed_k, False, graphed=False)
        for _ in range(20):
            step_k()
        k_ms, k_n = ctx.profile_read(0)
        ctx.profile_enable(False)
        res["fused_kernel_ms"] = k_ms / max(k_n, 1)
        res["around_fused_kernel_us_graph"] = (res["ms_per_step_graph"] - res["fused_kernel_ms"]) * 1e3
        res["around_fused_kernel_us_eager"] = (res["ms_per_step_eager"] - res["fused_kernel_ms"]) * 1e3
        res["projected_speedup_without_allreduce"] = ms_full / res["ms_per_step_graph"]
        res["projected_speedup_with_assumed_allreduce"] = ms_full / (res["ms_per_step_graph"] + assumed_allreduce_us[N] * 1e-3)
        out["ranks"][str(N)] = res
        neq_k.close(); rs_k.close()
    return out


def association_points(ctx, pv, torch, args, scans, dscans, ref, nei, associate, timed_assoc, make_step, F, ui, uj, dev):
    """Two more points SURVEY.md §8(d) asks for: raw targets (every one of the 65 536 points a surfLessFlat target, the
    wording of BASELINE.md §2) and the Floor plane tolerance 0.01."""
    out = {}
    # --- raw targets, the WHOLE batch (BASELINE.md §2's literal wording: every one of the 65 536 points of a scan is also a
    # surfLessFlat target): all scans re-uploaded with their full cloud as the target cloud, all pairs associated, and the
    # fused step run on what the reference's own accept tests let through
    S = args.scans
    sel = [(int(r), int(n)) for r, n in zip(ref, nei) if r < S and n < S]
    raw = {}
    for k in sorted(set(a for a, _ in sel) | set(b for _, b in sel)):
        d = dict(scans[k]); d["less_xyz"] = scans[k]["flat_xyz"]; d["less_tag"] = scans[k]["flat_tag"]
        raw[k] = pv.Scan(ctx, d)
    rr = np.array([a for a, _ in sel]); nn_ = np.array([b for _, b in sel])
    r0 = associate(rr, nn_, args.tolerance, raw); r0.close()                   # sizes the pool for this shape
    rsr, wall, ms, launches, allocs = timed_assoc(rr, nn_, args.tolerance, raw)
    nq = int(sum(len(scans[b]["flat_xyz"]) for _, b in sel)); nt = int(sum(len(scans[a]["flat_xyz"]) for a, _ in sel))
    out["raw_targets"] = {"pairs": len(sel), "queries": nq, "targets_per_pair": int(len(scans[0]["flat_xyz"])), "accepted": int(rsr.n),
                          "kernel_ms": ms, "wall_s": wall, "M_queries_per_s_kernel": nq / max(ms, 1e-9) / 1e3,
                          "compulsory_bytes": (nq + nt) * 16, "compulsory_GBps_at_kernel_time": (nq + nt) * 16 / max(ms, 1e-9) / 1e6,
                          "note": "10-NN of a query among raw VLP-16 returns lie on one ring: the reference's own collinearity test "
                                  "(FormLine(points, 3.0)) rejects most queries"}
    neq_r = pv.NormalEq(ctx, F, ui, uj)
    packed_r = torch.zeros(neq_r.size, dtype=torch.float64, device=dev)
    step_r, _ = make_step(rsr, neq_r, packed_r, False, graphed=False)
    for _ in range(20):
        step_r()
    torch.cuda.synchronize()
    reps = 100
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        step_r()
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / reps * 1e3
    k_ms, k_n = ctx.profile_read(0)
    ctx.profile_enable(False)
    k_avg = k_ms / max(k_n, 1)
    out["raw_targets"]["fused"] = {"what": "the fused step (r + 1x12 J -> per-pose blocks) on the blocks of the literal BASELINE.md §2 workload: %d pairs, raw 65 536-point "
                                           "targets" % len(sel), "evals": int(rsr.n), "fused_kernel_ms": k_avg, "ms_per_step": step_ms,
                                   "M_evals_per_s_kernel": rsr.n / max(k_avg, 1e-9) / 1e3, "M_evals_per_s_step": rsr.n / max(step_ms, 1e-9) / 1e3,
                                   "GBps_kernel": rsr.n * 56 / max(k_avg, 1e-9) / 1e6, "frac_of_hbm_peak": rsr.n * 56 / max(k_avg, 1e-9) / 1e6 / HBM_PEAK_GBPS,
                                   "evals_per_pair": rsr.n / max(len(sel), 1)}
    neq_r.close(); rsr.close()
    for d in raw.values():
        d.close()
    # --- Floor: lidar_plane_tolerance 0.01 (config/Floor.txt) on the first 256 reference scans, association + fused step
    keep = np.flatnonzero(ref < min(256, args.scans))
    rf, nf = ref[keep], nei[keep]
    r0 = associate(rf, nf, 0.01); r0.close()
    rsf, wall, ms, launches, allocs = timed_assoc(rf, nf, 0.01)
    nq = int(sum(len(scans[int(b)]["flat_xyz"]) for b in nf))
    neq_f = pv.NormalEq(ctx, F, ui, uj)
    packed_f = torch.zeros(neq_f.size, dtype=torch.float64, device=dev)
    step_f, _ = make_step(rsf, neq_f, packed_f, False, graphed=False)
    for _ in range(5):
        step_f()
    ctx.profile_enable(True)
    for _ in range(10):
        step_f()
    k_ms, k_n = ctx.profile_read(0)
    ctx.profile_enable(False)
    out["floor_tolerance_0.01"] = {"pairs": int(len(rf)), "queries": nq, "accepted": int(rsf.n), "accept_ratio": rsf.n / max(nq, 1),
                                   "assoc_kernel_ms": ms, "assoc_wall_s": wall, "M_queries_per_s_kernel": nq / max(ms, 1e-9) / 1e3,
                                   "fused_kernel_ms": k_ms / max(k_n, 1), "M_evals_per_s_kernel": rsf.n / max(k_ms / max(k_n, 1), 1e-9) / 1e3,
                                   "GBps": rsf.n * 56 / max(k_ms / max(k_n, 1), 1e-9) / 1e6}
    neq_f.close(); rsf.close()
    return out


def pcie_block(ctx, pv, associate, ref, nei, args):
    """Host-visible evaluation for a host-side solver (integration/pvlm_ceres.hpp): the blocks of the first 64 reference scans,
    evaluated on the GPU and delivered into page-locked host memory.  Two encodings: the 1 x 12 Jacobian rows (104 B per
    block) and wrench rows [r | c | g] + per-pair 3x3 tables (56 B per block, the row is formed on the host on demand)."""
    keep = np.flatnonzero(ref < min(64, args.scans))
    rs = associate(ref[keep], nei[keep], args.tolerance)
    n, P = rs.n, rs.n_pairs
    out = {"blocks": int(n), "pairs": int(P)}
    t0 = time.perf_counter()
    h_r = ctx.host_alloc(n * 8); h_J = ctx.host_alloc(n * 96); h_w = ctx.host_alloc(n * 56); h_t = ctx.host_alloc(max(P, 1) * 33 * 8); h_f = ctx.host_alloc(n * 32)
    out["pinned_alloc_s"] = time.perf_counter() - t0
    for name, fn, nbytes in (("jacobian_rows", lambda: rs.eval_host_async(h_r, h_J), 104), ("wrench_rows", lambda: rs.eval_wrench_host_async(h_w, h_t), 56),
                             ("force_rows", lambda: rs.eval_force_host_async(h_f, h_t), 32)):
        fn(); ctx.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"bytes_per_block": nbytes, "seconds": dt, "M_evals_per_s_host_visible": n / dt / 1e6, "D2H_GBps": n * nbytes / dt / 1e9}
    # pvlm_eval into ordinary (pageable) numpy arrays, on a slice: round 1 copied straight into them (94 M eval/s), now the
    # copy is staged through the context's pinned arena
    m = min(n, 2_000_000)
    keep2 = np.flatnonzero(ref < 4)
    rs2 = associate(ref[keep2], nei[keep2], args.tolerance)
    t0 = time.perf_counter()
    rs2.eval(jac=True)
    dt = time.perf_counter() - t0
    out["pageable_pvlm_eval"] = {"blocks": int(rs2.n), "M_evals_per_s_host_visible": rs2.n / dt / 1e6, "D2H_GBps": rs2.n * 104 / dt / 1e9}
    out["link_GBps_measured"] = max(out["jacobian_rows"]["D2H_GBps"], out["wrench_rows"]["D2H_GBps"])   # the boxes of the pool differ (30 - 56 GB/s seen)
    rs2.close(); rs.close()
    out["what_the_adapter_uses"] = "force_rows [r | g] for the point functors (integration/pvlm_ceres.hpp: the moment is rebuilt from the block's point in Evaluate), wrench_rows for plane / IOU blocks"
    for a in (h_r, h_J, h_w, h_t, h_f):
        ctx.host_free(a)
    return out


def pose_solve_block(ctx, scans=1593, reps=9):
    """K10: the reduced pose system of a Floor-sized EstimatePose (1593 poses, the proximity graph of tools/spd_floor_bench.py: 9 552 unknowns, 17 361 blocks) solved
    by pvlm_spd_solve_blocks — assembly on the device, nested-dissection factorisation as the tasks of ONE launch, both substitutions.  Wall per solve (median), with
    the plan cached as it is from the second LM step of a Solve on.  Upstream this is inside ceres::Solve (SPARSE_SCHUR, util/Optimization.cpp:638-666)."""
    from tools.spd_floor_bench import neighbours
    rng = np.random.default_rng(3)
    pairs = set()
    for i, nb in enumerate(neighbours(scans)):
        for j in nb:
            if i != j:
                pairs.add((min(i, j), max(i, j)))
    pairs = [(p, p) for p in range(scans)] + sorted(pairs)
    n = 6 * (scans - 1)
    off = np.arange(-6, n).reshape(scans, 6); off[0] = -1
    rows = np.array([off[pa] for pa, pb in pairs], np.int32); cols = np.array([off[pb] for pa, pb in pairs], np.int32)
    mirror = np.array([int(pa != pb) for pa, pb in pairs], np.int32)
    blocks = np.empty((len(pairs), 36))
    for k, (pa, pb) in enumerate(pairs):
        if pa == pb:
            J = rng.normal(size=(9, 6)); blocks[k] = (J.T @ J + 30 * np.eye(6)).reshape(-1)
        else:
            blocks[k] = (rng.normal(size=(6, 6)) * 0.2).reshape(-1)
    scale = np.full(n, 0.2); diag = np.full(n, 1.0); rhs = rng.normal(size=n)
    t0 = time.perf_counter()
    x, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    first = time.perf_counter() - t0
    walls = []
    for _ in range(reps):
        t0 = time.perf_counter()
        x, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
        walls.append(time.perf_counter() - t0)
    plan = ctx.spd_plan()
    return {"poses": scans, "unknowns": n, "blocks": len(pairs), "ms_per_solve": float(np.median(walls) * 1e3), "ms_first_call_with_plan": first * 1e3, "info": int(info),
            "plan": plan, "form": "one launch (k_nd_flow + k_nd_flow_bwd)" if plan.get("launched_levels") == 0 and plan.get("levels", 0) > 0 else "level launches",
            "solves_redone_with_level_launches": int(ctx.spd_one_launch()), "ms_per_solve_earlier": {"round 5, column by column": 11.2, "level launches": 7.1}}


def mvs_block(ctx, pv):
    """BASELINE.json config 5: the panoramic PatchMatch kernels on views RESIDENT in HBM (pvlm_mvs_views_*), at the reference's
    Room size (5.7K at scale -2 = 1440 x 720, config/Room.txt:87) and at the full 5.7K size.  Scene: a textured sphere of
    radius 3 m seen from three camera centres 0.2 m apart (every window projects into every neighbour).  K11 = scoring pass
    (InitConfMap) and K13 = one checkerboard PatchMatch iteration (two colour passes), both with one pixel per thread since round 3;
    K13s = one iteration of the sequential sweep the Room / Floor configs select (one launch per anti-diagonal, one wave per pixel),
    K12 = FilterDepthImageRefine.
    The NCC sums run in the reference's sequential order and the kernels are VALU-bound, so their roof is instruction
    issue, not HBM: the algorithmic HBM bytes (29 B per pixel and view touched) are reported next to the time."""
    out = {}
    for rows, cols in ((720, 1440), (2880, 5760)):
        out["%dx%d" % (cols, rows)] = mvs_one_size(ctx, rows, cols)
    try:
        out["k13s_sequential_batch_1440x720"] = mvs_batch_point(ctx)
    except Exception as e:                                   # e.g. a small pool reservation: the per-view numbers above stand on their own
        out["k13s_sequential_batch_1440x720"] = {"error": str(e)[:200]}
    out["bound"] = "VALU: the kernels' roof is instruction issue, the HBM fraction is reported for completeness"
    # SQ counters of the same kernels (tools/prof_r4_final.sh -> profiles/r4_pmc_mvs.json, separate --pmc pass of
    # tools/mvs_bench.py): wave VALU instructions x 4 cycles / (1024 SIMDs x kernel time) = a lower bound of the VALU pipes' load
    try:
        import glob
        src = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_mvs.json")))[-1]
        pmc = json.load(open(src))
        out["pmc"] = {"source": "profiles/" + os.path.basename(src),
                      "kernels": {k: {f: v[f] for f in ("simd_valu_util_lower_bound", "valu_issue_frac", "wait_frac", "valu_insts_per_wave", "kernel_trace_ms") if f in v}
                                  for k, v in pmc.items() if "k_mvs_conf" in k or "k_mvs_propagate" in k}}
    except Exception:
        out["pmc"] = None
    # counter-based roof: wave VALU instructions per pixel (SQ_INSTS_VALU of the pass above / pixels) against the rate a SIMD issues them at
    # (one wave64 instruction per 4 cycles, 1024 SIMDs x 2.4 GHz); `frac` = achieved pixels/s over the pixels/s at full issue
    try:
        issue = PEAKS["valu_wave_insts_per_s"]
        kernels = {}
        small = out.get("1440x720") or {}
        # algorithmic floor per pixel, in the counters' unit (wave instructions per pixel = operations of one pixel's lane / 64), counted from the per-pixel program the
        # reference prescribes (ScorePixel / ProcessPixel, mvs/MVS.cpp:774-923, :721-772; csrc/pvlm_mvs_core.h) — an estimate good to a few tens of per cent:
        # a hypothesis scored against one neighbour view = the plane-induced homography (30) + 49 taps (window 7 x 7, step 1: this block's configuration) of
        # [projection of the tap into the neighbour panorama with FastAtan2 (~60 as single-precision operations), four bilinear weights + blend (12), bilateral weight
        # + the five NCC sums (14)] + the NCC itself (25); K11 scores one hypothesis per pixel against two neighbours, a K13 iteration up to 4 propagated + 12
        # perturbed ones (config/Room.txt)
        per_hyp_view = 30 + 49 * (60 + 12 + 14) + 25
        floors = {"k_mvs_conf_lane": (2 * per_hyp_view + 40) / 64.0, "k_mvs_propagate_lane": (16 * 2 * per_hyp_view + 16 * 60) / 64.0}
        for kname, block in (("k_mvs_conf_lane", "k11_scoring_pass"), ("k_mvs_propagate_lane", "k13_patchmatch_iteration")):
            hit = [v for k, v in pmc.items() if kname in k and v.get("valu_wave_insts_per_unit")]
            if hit and block in small:
                per_pixel = hit[0]["valu_wave_insts_per_unit"]                         # wave instructions per pixel (a wave holds 64 pixels)
                kernels[kname] = roof("VALU issue", small[block]["M_pixels_per_s"], "M pixels/s", issue / per_pixel / 1e6, issue / floors[kname] / 1e6,
                                      valu_wave_insts_per_pixel=per_pixel, algorithmic_floor_valu_per_pixel=floors[kname])
        roof_ = {"bound": "VALU issue", "wave_insts_per_s_peak": issue, "kernels": kernels}
        out["roof"] = roof_
    except Exception:
        out["roof"] = None
    return out


def mvs_one_size(ctx, rows, cols, k13_reps=2):
    """One reference view against two resident neighbours at rows x cols: K11, one K13 iteration, K12 (see mvs_block)."""
    from panovlm_amd.api import MvsViews
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    lon = (2 * xx / cols - 1) * np.pi; lat = (0.5 - yy / rows) * np.pi
    ray = np.stack([np.cos(lat) * np.sin(lon), -np.sin(lat), np.cos(lat) * np.cos(lon)], axis=-1).astype(np.float32)
    gray = np.clip(128 + 50 * np.sin(40 * lon) * np.cos(37 * lat) + 40 * np.sin(91 * lat + 13 * lon), 0, 255).astype(np.uint8)
    depth = np.full((rows, cols), 3.0, np.float32)
    normal = (-ray).astype(np.float32)
    V = MvsViews(ctx, rows, cols, 3)
    for v in range(3):
        V.upload(v, gray=gray, depth=depth, normal=normal, conf=np.zeros((rows, cols), np.float32))
        V.snapshot_depth(v)
    Rn = np.stack([np.eye(3, dtype=np.float32)] * 2); tn = np.array([[0.2, 0, 0], [-0.2, 0.01, 0.05]], np.float32)
    ref, nei = 0, [1, 2]
    res = {"pixels": rows * cols, "neighbours": 2, "window": "7x7"}
    for name, fn, reps in (("k11_scoring_pass", lambda: V.estimate(ref, nei, Rn, tn, max_iter=-1), 3),
                           ("k13_patchmatch_iteration", lambda: V.estimate(ref, nei, Rn, tn, max_iter=1, seed=3), k13_reps),
                           ("k13s_sequential_iteration", lambda: V.estimate(ref, nei, Rn, tn, max_iter=1, seed=3, sequential=True), 1),
                           ("k12_fusion_filter_refine", lambda: V.filter_refine(ref, nei, Rn, tn), 3)):
        fn(); ctx.synchronize()
        V.upload(ref, depth=depth, normal=normal, conf=np.zeros((rows, cols), np.float32))
        if name.startswith("k13"):
            V.estimate(ref, nei, Rn, tn, max_iter=-1)
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ms = ctx.timer_stop() / reps
        texels = rows * cols * 49 * 2
        res[name] = {"ms": ms, "M_pixels_per_s": rows * cols / ms / 1e3}
        if name.startswith("k11"):
            res[name]["G_texel_projections_per_s"] = texels / ms / 1e6
        res[name]["algorithmic_hbm_bytes"] = rows * cols * 29 * 3
        res[name]["frac_of_hbm_peak"] = rows * cols * 29 * 3 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
    V.close()
    return res


def mvs_batch_point(ctx, rows=720, cols=1440, views=64):
    """The sweep the Room / Floor configs select (sequential), as a pipeline over many frames runs it: `views` resident reference views per
    launch (pvlm_mvs_views_estimate_sequential_batch; upstream: one image per OpenMP thread), two neighbours each — ms per view and
    iteration.  From 32 768 pixels per anti-diagonal the launch uses four threads per pixel, below that one wave per pixel."""
    from panovlm_amd.api import MvsViews
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    lon = (2 * xx / cols - 1) * np.pi; lat = (0.5 - yy / rows) * np.pi
    ray = np.stack([np.cos(lat) * np.sin(lon), -np.sin(lat), np.cos(lat) * np.cos(lon)], axis=-1).astype(np.float32)
    gray = np.clip(128 + 50 * np.sin(40 * lon) * np.cos(37 * lat) + 40 * np.sin(91 * lat + 13 * lon), 0, 255).astype(np.uint8)
    depth = np.full((rows, cols), 3.0, np.float32)
    normal = (-ray).astype(np.float32)
    V = MvsViews(ctx, rows, cols, views + 2)
    for v in range(views + 2):
        V.upload(v, gray=gray, depth=depth, normal=normal, conf=np.zeros((rows, cols), np.float32))
    Rn = np.stack([np.eye(3, dtype=np.float32)] * 2); tn = np.array([[0.2, 0, 0], [-0.2, 0.01, 0.05]], np.float32)
    jobs = [dict(ref=k, nei=[views, views + 1], R_nr=Rn, t_nr=tn, seed=3 + k) for k in range(views)]
    for k in range(views):
        V.estimate(k, [views, views + 1], Rn, tn, max_iter=-1)
    ctx.synchronize()
    ctx.timer_start()
    V.estimate_sequential_batch(jobs, max_iter=1)
    ms = ctx.timer_stop()
    V.close()
    return {"rows": rows, "cols": cols, "views_per_launch": views, "neighbours": 2, "ms_per_iteration": ms, "ms_per_view_and_iteration": ms / views,
            "M_pixels_per_s": views * rows * cols / ms / 1e3}


def features_block(ctx, pv, scans=454, cols=1800):
    """The LiDAR feature extractor of a Room-sized batch (454 raw scans of 16 x 1800, firing order, ~28.5 k returns each) in one
    pvlm_ring_extract_batch_picks — range image, segmentation, curvature, the sector orders (K23), the edge / plane picks and the voxel grid (K24):
    HIP-event milliseconds per stage, wall per batch, and the oracle's statement-by-statement loop on one host core beside it."""
    from panovlm_amd import synthetic as sy
    base = [sy.raw_vlp16_scan(k, cols=cols, clutter=40) for k in range(8)]
    raws = [base[k % len(base)] for k in range(scans)]
    pv.RingBatch(ctx, raws[:8], n_rings=16, horizon=cols).close()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        b = pv.RingBatch(ctx, raws, n_rings=16, horizon=cols, segment=True, picks=(1000.0, 5.0), keep_arrays=False)      # what the C++ host mirror asks for
        wall = (time.perf_counter() - t0) * 1e3
        tm = b.timing()
        if best is None or wall < best[0]:
            sample = [b.picks(k) for k in range(0, scans, 16)]
            best = (wall, tm, sum(b.result(k)["resolved_points"] for k in range(0, scans, 16)), b.result(0)["n_kept"], b.result(0)["n_reordered"],
                    sum(int(q["ring_host"].any()) for q in sample), len(sample), sum(len(q["less_flat"]) for q in sample) / len(sample),
                    sum(len(q["corner"]) for q in sample) / len(sample))
        b.close()
    wall, tm, resolved, kept, reordered, undecided, sampled, centroids, corners = best
    device_ms = sum(v for k, v in tm.items() if k not in ("upload", "download"))
    out = {"scans": scans, "rings_x_columns": [16, cols], "points": int(sum(len(r) for r in raws)), "wall_ms_per_batch": wall, "ms_per_scan_wall": wall / scans,
           "device_ms_per_batch": device_ms, "stage_ms": tm, "copies_ms": tm["upload"] + tm["download"],
           "points_decided_by_host_libm_sampled": int(resolved), "kept_of_reordered_scan0": [int(kept), int(reordered)],
           "scans_with_a_ring_left_to_the_host_sampled": [int(undecided), int(sampled)], "edge_picks_per_scan": corners, "less_flat_centroids_per_scan": centroids,
           "what": "ReOrderVLP + Segmentation + adaptive curvature + sector sort (std::sort's order of equal curvatures restated) + ExtractEdgeFeatures2 / "
                   "ExtractPlaneFeatures2 picks + pcl::VoxelGrid of the less-flat points (sensors/Velodyne.cpp:371-526, :1438-1586, :623-657, :883-1000, :1098-1189); "
                   "compact_curvature = K21 + K22 + K23 + K24; bit-exact vs the oracle: tests/test_ring_gpu.py.  The line branch (EdgeToLine: growth on the GPU, K27; fusion and "
                   "filters on the host) is in extract_features_batch below"}
    # roof of the batch: the host link.  The boundary hands over host buffers and takes host arrays back: 16 B per raw point up; down — since round 6 — source index +
    # (ring, column) of every kept point (8 B), 1 B of state and ~2 B of pick lists and centroids; the other five per-point arrays (16 B: curvature, window, range,
    # sector order) only for the scans with a ring K24 left to the host (round 5: always, 43 B per point in all)
    link = PEAKS["host_link_GBps"] * 1e9
    bytes_per_point = 16 + 8 + 1 + 2 + 16.0 * undecided / max(sampled, 1)
    moved = out["points"] * bytes_per_point
    out["roof"] = {"bound": "host link (PCIe Gen5 x16)", "bytes_per_point": bytes_per_point, "bytes_per_point_round5": 43, "bytes_per_batch": moved, "GBps_assumed": link / 1e9,
                   "ms_at_link_rate": moved / link * 1e3, "frac": (moved / link * 1e3) / wall, "device_ms_share": device_ms / wall}
    try:
        from oracle import oracle as orc
        t0 = time.perf_counter()
        for k in range(3):
            orc.ScanFeatures(raws[k], n_scans=16, horizon=cols, extract=False)
        out["cpu_oracle_ms_per_scan_reorder"] = (time.perf_counter() - t0) / 3 * 1e3
        t0 = time.perf_counter()
        for k in range(3):
            orc.ScanFeatures(raws[k], n_scans=16, horizon=cols, segment=True, extract=True)
        out["cpu_oracle_ms_per_scan_full_extraction"] = (time.perf_counter() - t0) / 3 * 1e3
    except Exception as e:
        out["cpu_oracle_error"] = str(e)[:120]
    return out


def extract_features_batch_block(scans=454, threads=32):
    """The whole Velodyne::ExtractFeaturesBatch of the host mirror (libpvlm_host.so through its test driver, its own process and context): the device batch
    above + the line branch (growth on the GPU, K27; fusion and filters on the host) and the assembly of the clouds on host threads, for a Room-sized batch of raw scans.  Wall of the best of three calls (each starts
    after a pause: the boxes of this pool cap a process at 16 CPUs per 100 ms) and the thread-milliseconds of the host stages."""
    import re
    import subprocess
    import tempfile
    from panovlm_amd import synthetic as sy
    from tests import host_io
    base = []
    for k in range(16):
        R, t = sy.estimated_pose(k)
        base.append(dict(id=k, R_wl=R, t_wl=t, raw=sy.raw_vlp16_scan(k, clutter=40)))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "raw.bin")
        host_io.write_raw_scans(path, [dict(base[k % 16], id=k) for k in range(scans)])
        env = dict(os.environ, PVLM_FEATURE_PROFILE="1")
        r = subprocess.run([host_io.driver(), "featbench_gpu", path, "3", "1", str(threads)], capture_output=True, text=True, timeout=600, env=env)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-300:])
    walls = [float(m.group(1)) for m in re.finditer(r"featbench_gpu .* wall_ms ([0-9.]+)", r.stdout)]
    prof = [l for l in r.stderr.splitlines() if l.startswith("feature_profile ExtractFeaturesBatch")]
    stages = {}
    if prof:
        for name, ms in re.findall(r"\[([^\]]+)\] ([0-9.]+)", prof[-1]):
            stages[name] = float(ms)
    grow = {}
    k27 = [l for l in r.stderr.splitlines() if l.startswith("feature_profile line growth on the GPU (K27)")]
    if k27:
        m = re.search(r"([0-9.]+) ms on the producer \(kernels ([0-9.]+) ms, ([0-9]+) tasks\); scans the device handed back to the host growth: ([0-9]+)", k27[-1])
        if m:
            grow = {"producer_ms": float(m.group(1)), "kernel_ms": float(m.group(2)), "tasks": int(m.group(3)), "scans_handed_back_to_the_host": int(m.group(4)),
                    "what": "K27 (pvlm_line_grow_begin / _finish): every (start point, neighbour pair) of ExtractLineFeatures grown as a task of its own, a lane each, in rounds "
                            "of speculative start points with upstream's walk replayed on the device; the growth of device batch k runs beside the range-image stages of "
                            "batch k + 1.  Round 5 grew on the host threads: EdgeToLine 574 thread-ms per call"}
    return {"scans": scans, "host_threads": threads, "wall_ms_per_call": min(walls[1:] or walls), "wall_ms_all_calls": walls, "host_thread_ms_last_call": stages,
            "line_growth_on_the_gpu": grow, "cpu_quota": cpu_quota(),
            "what": "Velodyne::ExtractFeaturesBatch (raw scans -> cloud_scan, cornerSharp / cornerLessSharp, surfFlat / surfLessFlat, line segments): range image, "
                    "sector orders, picks, voxel grid and the growth of the line segments on the GPU in two overlapped device batches; the fusion and the filters of the "
                    "line branch and the assembly of the clouds on the host threads; first call includes code-object loading and pinned allocations.  Round 4: 69.7 ms, "
                    "round 5: 33.9 ms with 574 host thread-ms of EdgeToLine.  Every output equal to the oracle: tests/test_host_gpu.py, tests/test_linegrow_gpu.py"}


def undistort_block(ctx, pv, scans=454, cols=1800):
    """Motion compensation of a Room-sized batch (pvlm_undistort_batch, K25: Velodyne::UndistortCloud of every scan of LidarOdometry::UndistortLidars in one call):
    wall per call with the clouds on the host, as the boundary has them — 16 B up and 16 B down per point over the host link — and the oracle's per-point loop on
    one host core beside it."""
    from panovlm_amd import api, synthetic as sy
    base = [sy.raw_vlp16_scan(k, cols=cols, clutter=40) for k in range(8)]
    clouds = [base[k % len(base)] for k in range(scans)]
    starts, ends = [], []
    for k in range(scans):
        R0, t0 = sy.estimated_pose(k % 16); R1, t1 = sy.estimated_pose(k % 16 + 1)
        starts.append((R0, t0)); ends.append((R1, t1))
    api.undistort_batch(ctx, clouds[:8], starts[:8], ends[:8])
    best = None
    for _ in range(3):
        work = [c.copy() for c in clouds]                     # the C call works in place, as upstream's loop does on lidars[i].cloud
        t0 = time.perf_counter()
        api.undistort_batch(ctx, work, starts, ends, inplace=True)
        w = (time.perf_counter() - t0) * 1e3
        best = w if best is None or w < best else best
    points = int(sum(len(c) for c in clouds))
    link = PEAKS["host_link_GBps"] * 1e9
    out = {"scans": scans, "points": points, "wall_ms_per_call": best, "M_points_per_s": points / best / 1e3,
           "includes": "descriptor marshalling of the Python wrapper, staging copies into / out of pinned memory (host threads), both link directions, the kernel",
           "roof": {"bound": "host link (PCIe Gen5 x16), both directions in sequence", "bytes_per_point": 32, "ms_at_link_rate": points * 32 / link * 1e3,
                    "frac": (points * 32 / link * 1e3) / best},
           "what": "Velodyne::UndistortCloud (sensors/Velodyne.cpp:1642-1674) per point: slerp of the end-to-start rotation by i / n (two double sines), rotation, "
                   "translation; bit-identical to the oracle on the test clouds (tests/test_undistort_gpu.py, tolerance 1e-6)"}
    try:
        from oracle import oracle as orc
        t0 = time.perf_counter()
        for k in range(3):
            orc.undistort_cloud(clouds[k], *starts[k], *ends[k])
        out["cpu_oracle_ms_per_scan"] = (time.perf_counter() - t0) / 3 * 1e3
    except Exception as e:
        out["cpu_oracle_error"] = str(e)[:120]
    return out


def panorama_block(ctx, pv, torch, dev, with_votes=True):
    rows, cols = 2880, 5760
    n = rows * cols
    g = torch.Generator(device=dev); g.manual_seed(5)
    cam = torch.randn((n, 3), generator=g, device=dev, dtype=torch.float32) * 3.0
    px = torch.empty((n, 2), device=dev, dtype=torch.float32)
    out = {"rows": rows, "cols": cols, "pixels": n}
    reps = 10
    for name, fn, nbytes in (("cam_to_image_f32", lambda: ctx.cam_to_image_f32_dev(rows, cols, n, cam.data_ptr(), px.data_ptr()), 20),
                             ("image_to_cam_f32", lambda: ctx.image_to_cam_f32_dev(rows, cols, n, px.data_ptr(), 1.0, cam.data_ptr()), 20)):
        fn(); ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ms = ctx.timer_stop() / reps
        out[name] = {"kernel_ms": ms, "G_points_per_s": n / ms / 1e6, "GBps": n * nbytes / ms / 1e6, "bytes_per_point": nbytes,
                     "frac_of_hbm_peak": n * nbytes / ms / 1e6 / HBM_PEAK_GBPS}
    if not with_votes:
        return out
    # camera<->LiDAR voting: 454 frames x 3 neighbouring scans (Room, neighbor_size_joint = 3), 200 image lines per
    # panorama, 1500 corner points in 40 segments per scan
    rng = np.random.default_rng(11)
    from panovlm_amd import synthetic as sy
    lw = sy.random_world_lines(rng, 40, extent=4.0)
    scan = sy.make_line_scan(rng, 0, np.eye(3), np.zeros(3), lw, pts_per_line=(30, 45), extra_pts=60)
    local = dict(scan); local["corner_xyz"] = scan["corner_local"]
    dscan = pv.Scan(ctx, local)
    lines = rng.uniform([0, 0, 0, 0], [cols, rows, cols, rows], size=(200, 4)).astype(np.float32)
    pairs = 454 * 3
    T = np.eye(4)
    ctx.cam_lidar_votes_batch_sparse(rows, cols, [lines] * 8, [dscan] * 8, [T] * 8)       # code objects loaded, staging sized
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    voff, nz_index, nz_count = ctx.cam_lidar_votes_batch_sparse(rows, cols, [lines] * pairs, [dscan] * pairs, [T] * pairs)
    wall = time.perf_counter() - t0
    k8_ms, k8_n = ctx.profile_read(3)
    ctx.profile_enable(False)
    tests_n = pairs * len(lines) * len(scan["corner_local"])
    # K8 has no HBM term (1500 points x 200 lines per pair are re-read from L2): its roof is VALU issue.  One wave64 VALU instruction occupies its
    # SIMD for 4 cycles (measured on these kernels, profiles/r5_assoc_variants.txt: 4.3-4.5 cycles per instruction at >= 90 % busy) -> 1024 SIMDs x
    # 2.4 GHz / 4 = 614 G wave instructions per second at most; `valu_per_test` and `simd_valu_busy` come from the SQ counters of the kernel
    # (rocprofv3 --pmc pass of this block: SQ_INSTS_VALU / tests, SQ_INSTS_VALU x 4 / (1024 x kernel time x 2.4 GHz)), not from its source.
    pmc = {}
    try:
        src = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_k8.json")))[-1]
        pmc = json.load(open(src)); pmc["source"] = "profiles/" + os.path.basename(src)
        if pmc.get("kernel") != "k_cam_lidar_votes_points":       # counters of the thread-per-test kernel of rounds 2-5: not this kernel's instruction stream
            pmc = {"source": pmc["source"], "note": "counters of another kernel (%s): no roof derived from them" % pmc.get("kernel")}
    except Exception:
        pmc = {"source": None}
    kernel_ms = k8_ms / max(k8_n, 1)
    roof_k8 = None
    if pmc.get("valu_wave_insts_per_test"):
        peak = PEAKS["valu_wave_insts_per_s"] / pmc["valu_wave_insts_per_test"] / 1e9
        # (the kernel since round 6: a thread per point walking the pair's lines, k_cam_lidar_votes_points)
        # algorithmic floor of one (line, point) test (joint_optimization/CameraLidarLineAssociate.cpp:394-426): the transformed point is shared by the lines of a
        # pair; per test two plane-side dot products (2 x 5), the range comparison and two angle comparisons against precomputed cosines (2 x 2) + the vote = ~16
        # operations over the 64 lanes of a wave -> 0.25 wave instructions per test at one test per lane
        floor_wave_insts_per_test = 16.0 / 64.0
        roof_k8 = roof("VALU issue: 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction", tests_n / max(k8_ms, 1e-9) / 1e6, "G tests/s", peak,
                       PEAKS["valu_wave_insts_per_s"] / floor_wave_insts_per_test / 1e9, valu_wave_insts_per_test=pmc["valu_wave_insts_per_test"],
                       algorithmic_floor_wave_insts_per_test=floor_wave_insts_per_test, G_tests_per_s=peak, counters=pmc)
    out["cam_lidar_votes"] = {"pairs": pairs, "image_lines": len(lines), "corner_points": int(len(scan["corner_local"])), "segments": int(dscan.n_segments),
                              "point_line_tests": int(tests_n), "wall_s_incl_copies": wall, "G_tests_per_s_incl_copies": tests_n / wall / 1e9,
                              "c_abi_call_s": getattr(ctx, "last_call_s", None), "call_over_kernel": getattr(ctx, "last_call_s", 0.0) * 1e3 / max(k8_ms, 1e-9),
                              "kernel_ms": kernel_ms, "G_tests_per_s_kernel": tests_n / max(k8_ms, 1e-9) / 1e6,
                              "roof": roof_k8,
                              "wall_over_kernel": wall * 1e3 / max(k8_ms, 1e-9),
                              "wall_is": "host line tables (272 400 rows, 16 host threads) + descriptors + the voting kernel + two small kernels that compact the "
                                         "non-zero counters + their read-back (pvlm_cam_lidar_votes_batch_sparse); round 4 copied 43 MB of dense blocks back",
                              "dense_counters": int(voff[-1]), "nonzero_counters": int(len(nz_index)), "votes_cast": int(nz_count.sum())}
    dscan.close()
    return out


def cpu_baseline(ctx, pv, dscans, ref, nei, aa, t, kind, args):
    """The CPU oracle (restated reference algorithm: Jet<12> AutoDiff through the reference's rotation
    chain, base/CostFunction.h) timed on this box's host cores on a bounded sample of the same batch:
    the residual blocks of the first pairs, all host threads (Ceres runs num_threads = 25 upstream)."""
    from oracle import oracle as orc
    quota = cpu_quota()
    threads = orc.num_threads() if not quota else max(1, min(orc.num_threads(), int(quota + 0.5)))     # threads beyond the cgroup's quota are throttled, not run
    npairs = min(len(ref), 64)
    small = ctx.assoc_point2plane([dscans[int(r)] for r in ref[:npairs]], [dscans[int(n)] for n in nei[:npairs]], args.tolerance, 1.0,
                                  kind=kind, flags=pv.FLAG_NORMALIZE_DISTANCE)
    off, rr, nn, rows = small.download()
    small.close()
    rid = np.repeat(rr, np.diff(off)).astype(np.int32); nid = np.repeat(nn, np.diff(off)).astype(np.int32)
    orows = np.concatenate([rows, np.ones((rows.shape[0], 1))], axis=1)
    n = rows.shape[0]
    orc.evaluate(kind, orows[:min(n, 50_000)], rid[:min(n, 50_000)], nid[:min(n, 50_000)], aa, t, normalize=True, jac=True, threads=threads)  # warm the thread pool
    # three samples of cpu_seconds each, median reported with the spread: on a 128-thread box shared with the driver the figure moved by
    # 30 % between rounds on unchanged code (4.42 -> 3.18 M eval/s) — it is a noisy number and the line says so
    samples = []
    reps = 0; dt = 0.0
    for _ in range(3):
        reps = 0
        t0 = time.perf_counter()
        while True:
            orc.evaluate(kind, orows, rid, nid, aa, t, normalize=True, jac=True, threads=threads)
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= args.cpu_seconds:
                break
        samples.append(n * reps / dt / 1e6)
    samples.sort()
    # SURVEY.md §8(d) also asks for the single-thread figure: a 3 s slice of the same rows on 1 thread
    n1 = min(n, 200_000); reps1 = 0
    t1 = time.perf_counter()
    while True:
        orc.evaluate(kind, orows[:n1], rid[:n1], nid[:n1], aa, t, normalize=True, jac=True, threads=1)
        reps1 += 1
        dt1 = time.perf_counter() - t1
        if dt1 >= min(3.0, args.cpu_seconds):
            break
    # ... and the unoptimised build: the reference's CMakeLists.txt sets no -O flag, so its PCL + Ceres path runs at -O0
    v_o0 = None
    try:
        n0 = min(n, 50_000); reps0 = 0
        t2 = time.perf_counter()
        while True:
            orc.evaluate_unoptimised(kind, orows[:n0], rid[:n0], nid[:n0], aa, t, normalize=True, threads=1)
            reps0 += 1
            dt2 = time.perf_counter() - t2
            if dt2 >= min(2.0, args.cpu_seconds):
                break
        v_o0 = n0 * reps0 / dt2 / 1e6
    except Exception:
        pass
    return {"value": samples[1], "unit": "M evals/s", "cores": threads, "kind": "port", "cpu_quota": quota, "hardware_threads": os.cpu_count(),
            "samples": samples, "spread": (samples[-1] - samples[0]) / max(samples[1], 1e-12), "noise_note": "median of 3 x %.0f s; min / max beside it" % args.cpu_seconds,
            "value_1thread": n1 * reps1 / dt1 / 1e6, "value_1thread_O0": v_o0,
            "sample": "%d residual blocks of the first %d pairs x %d repetitions; r + 1x12 J by Jet<12> AutoDiff (restated "
                      "reference algorithm, g++ -O2), OpenMP %d threads, 3 samples of %.1f s" % (n, npairs, reps, threads, dt)}


if __name__ == "__main__":
    main()
