"""Runtime around the kernels, on the GPU and through the C ABI: the context's device pool (a steady state makes no
hipMalloc), the streamed association (any batch size gives the same residual set, bit for bit), the binding of
pvlm_neq to a residual set that was destroyed and re-created, the HIP graph of an LM step, and the sharded path
executed by HIP kernels: two ranks sharing GPU 0, each associating + accumulating its shard of the pair list,
all-reduce, equal to the single-rank packed buffer (the loop that is sharded is
util/Optimization.cpp:521-560)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

try:        # torch bundles its own HIP runtime: when both are in one process it has to be loaded before libpvlm.so
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

from panovlm_amd import sharding as sh
from panovlm_amd import synthetic as sy

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = 0x100
LOSS_A = 2 * np.pi / 180


def _scans(F, cols=256):
    return {k: sy.make_scan(k, cols=cols, downsample_targets=0.2) for k in range(F)}


def _poses(F):
    ps = [sy.pose_params(*sy.estimated_pose(k)) for k in range(F)]
    return np.array([p[0] for p in ps]), np.array([p[1] for p in ps])


def _associate(ctx, pv, dev, ref, nei, flags=0):
    return ctx.assoc_point2plane([dev[int(r)] for r in ref], [dev[int(n)] for n in nei], 0.05, 1.0, kind=pv.POINT2PLANE_ANGLE,
                                 flags=pv.FLAG_NORMALIZE_DISTANCE | flags)


def test_pool_steady_state_and_streamed_batches_bit_exact():
    import panovlm_amd as pv
    F = 6
    scans = _scans(F)
    ref, nei = sy.pair_list(F, 4)
    ctx = pv.Context(0)
    dev = {k: pv.Scan(ctx, s) for k, s in scans.items()}
    rs = _associate(ctx, pv, dev, ref, nei, KEEP)
    off0, r0, n0, rows0 = rs.download()
    q0, nn0 = rs.assoc_debug()
    rs.close()
    a0 = ctx.mem_info()
    assert a0["reserved"] > 0 and a0["peak"] >= a0["in_use"]
    # second and third call: same sizes -> served entirely by the pool
    for _ in range(2):
        rs = _associate(ctx, pv, dev, ref, nei, KEEP)
        off1, r1, n1, rows1 = rs.download()
        assert np.array_equal(off0, off1) and np.array_equal(rows0, rows1)
        rs.close()
    a1 = ctx.mem_info()
    assert a1["device_allocs"] == a0["device_allocs"], (a0, a1)
    assert a1["in_use"] == a0["in_use"]
    for d in dev.values():
        d.close()
    ctx.trim()
    assert ctx.mem_info()["in_use"] <= a0["in_use"]
    ctx.close()
    # tiny batches: every pair its own batch, odd / even pipeline slots, many column blocks
    code = (
        "import numpy as np, sys; sys.path.insert(0, %r)\n"
        "import panovlm_amd as pv\nfrom panovlm_amd import synthetic as sy\n"
        "F = 6; scans = {k: sy.make_scan(k, cols=256, downsample_targets=0.2) for k in range(F)}\n"
        "ref, nei = sy.pair_list(F, 4)\n"
        "ctx = pv.Context(0); dev = {k: pv.Scan(ctx, s) for k, s in scans.items()}\n"
        "rs = ctx.assoc_point2plane([dev[int(r)] for r in ref], [dev[int(n)] for n in nei], 0.05, 1.0, kind=pv.POINT2PLANE_ANGLE, flags=pv.FLAG_NORMALIZE_DISTANCE | 0x100)\n"
        "off, r, n, rows = rs.download(); q, nn = rs.assoc_debug()\n"
        "aa = np.zeros((F, 3)); t = np.zeros((F, 3)); ctx.set_poses(aa, t)\n"
        "res, J = rs.eval(jac=True); blocks = rs.pair_blocks(pv.LOSS_HUBER, 0.03)\n"
        "np.savez(sys.argv[1], off=off, rows=rows, q=q, nn=nn, res=res, J=J, blocks=blocks)\n" % ROOT)
    out = {}
    for tag, rows_env in (("small", "5000"), ("default", None)):
        env = dict(os.environ)
        if rows_env:
            env["PVLM_ASSOC_BATCH_ROWS"] = rows_env
        path = "/tmp/pvlm_stream_%s_%d.npz" % (tag, os.getpid())
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=600)
        out[tag] = dict(np.load(path))
        os.remove(path)
    assert np.array_equal(out["default"]["off"], off0) and np.array_equal(out["default"]["rows"], rows0)
    assert np.array_equal(out["default"]["q"], q0) and np.array_equal(out["default"]["nn"], nn0)
    for k in ("off", "rows", "q", "nn", "res", "J", "blocks"):
        assert np.array_equal(out["small"][k], out["default"][k]), k


def test_neq_rebinds_when_a_set_is_recreated_at_the_same_address():
    """pvlm_neq cached its gather lists by the address of the residual set; a set destroyed and re-created
    by a re-association usually gets the old address back."""
    import panovlm_amd as pv
    from tests import synth
    rng = np.random.default_rng(5)
    F = 7
    aa, t = synth.random_poses(rng, F)
    ctx = pv.Context(0)
    ctx.set_poses(aa, t)
    pairs_a = (np.array([0, 1, 2, 3], np.int32), np.array([1, 2, 3, 4], np.int32))
    pairs_b = (np.array([4, 5, 6, 2], np.int32), np.array([5, 6, 4, 0], np.int32))   # same count, different structure
    ui, uj = sh.unordered_pairs(np.concatenate([pairs_a[0], pairs_b[0]]), np.concatenate([pairs_a[1], pairs_b[1]]))
    neq = pv.NormalEq(ctx, F, ui, uj)
    got, want = [], []
    for ref, nei in (pairs_a, pairs_b, pairs_a):
        rows, off = synth.random_resset(rng, pv.POINT2PLANE_ANGLE, aa, t, ref, nei, np.full(len(ref), 40))
        rs = pv.ResidualSet.upload(ctx, pv.POINT2PLANE_ANGLE, rows, off, ref, nei, flags=pv.FLAG_NORMALIZE_DISTANCE)
        got.append(neq.accumulate(rs, pv.LOSS_HUBER, LOSS_A))
        want.append(sh.pack_from_pair_blocks(rs.pair_blocks(pv.LOSS_HUBER, LOSS_A), ref, nei, F, ui, uj))
        rs.close()     # the next upload has the same sizes: the allocator hands the same addresses out again
    for g, w in zip(got, want):
        assert np.allclose(g, w, rtol=1e-12, atol=1e-12 * np.abs(w).max())
    assert not np.allclose(got[0], got[1])
    neq.close()
    ctx.close()


def test_graph_of_a_step_equals_eager_steps():
    import torch
    import panovlm_amd as pv
    F = 8
    scans = _scans(F)
    ref, nei = sy.pair_list(F, 4)
    aa, t = _poses(F)
    ui, uj = sh.unordered_pairs(ref, nei)
    dev_t = torch.device("cuda", 0)
    ctx = pv.Context(0)                       # own (capturable) stream
    dev = {k: pv.Scan(ctx, s) for k, s in scans.items()}
    rs = _associate(ctx, pv, dev, ref, nei)
    neq = pv.NormalEq(ctx, F, ui, uj)
    d_aa = torch.from_numpy(aa.copy()).to(dev_t); d_t = torch.from_numpy(t.copy()).to(dev_t)
    packed = torch.zeros(neq.size, dtype=torch.float64, device=dev_t)
    torch.cuda.synchronize()

    def step():
        ctx.set_poses_dev(F, d_aa.data_ptr(), d_t.data_ptr())
        neq.accumulate_dev(rs, packed.data_ptr(), pv.LOSS_HUBER, LOSS_A, zero_first=True)

    step(); ctx.synchronize()
    eager0 = packed.cpu().numpy().copy()
    ctx.graph_begin()
    step()
    g = ctx.graph_end()
    packed.zero_(); torch.cuda.synchronize()
    g.launch(); ctx.synchronize()
    assert np.array_equal(packed.cpu().numpy(), eager0)          # same kernels, same order: bit-identical
    # new parameter point: the graph reads the pose arrays it was captured with
    d_t.add_(0.01); d_aa.mul_(1.01); torch.cuda.synchronize()
    g.launch(); ctx.synchronize()
    replay = packed.cpu().numpy().copy()
    step(); ctx.synchronize()
    assert np.array_equal(packed.cpu().numpy(), replay)
    assert not np.array_equal(replay, eager0)
    # a capture that would have to allocate is refused, and the context stays usable
    neq2 = pv.NormalEq(ctx, F, ui, uj)
    ctx.graph_begin()
    with pytest.raises(pv.PvlmError):
        neq2.accumulate_dev(rs, packed.data_ptr(), pv.LOSS_HUBER, LOSS_A, zero_first=True)
    try:
        ctx.graph_end().close()
    except pv.PvlmError:
        pass
    step(); ctx.synchronize()
    assert np.array_equal(packed.cpu().numpy(), replay)
    # host-pointer entry points are not capturable (a staged copy would replay whatever the pinned arena holds by then; a
    # synchronisation invalidates the capture): refused with PVLM_ERR_STATE, the capture itself survives and replays correctly
    ctx.graph_begin()
    with pytest.raises(pv.PvlmError):
        ctx.set_poses(aa, t)
    with pytest.raises(pv.PvlmError):
        rs.eval(jac=False)
    with pytest.raises(pv.PvlmError):
        ctx.synchronize()
    step()
    g2 = ctx.graph_end()
    packed.zero_(); torch.cuda.synchronize()
    g2.launch(); ctx.synchronize()
    assert np.array_equal(packed.cpu().numpy(), replay)
    g2.close()
    g.close(); neq.close(); neq2.close(); rs.close()
    ctx.close()


# ---- two ranks, one GPU: the sharded path on HIP kernels ------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank_packed(rank, world, F, cols):
    """What one rank of bench.py does, with the exchange left to the caller: associate + accumulate its shard."""
    import torch
    import panovlm_amd as pv
    ref_all, nei_all = sy.pair_list(F, 4)
    ref, nei = sh.shard_pairs(ref_all, nei_all, F, rank, world)
    needed = sorted(set(ref.tolist()) | set(nei.tolist()))
    ctx = pv.Context(0)
    dev = {k: pv.Scan(ctx, sy.make_scan(k, cols=cols, downsample_targets=0.2)) for k in needed}
    rs = _associate(ctx, pv, dev, ref, nei)
    aa, t = _poses(F)
    ui, uj = sh.unordered_pairs(ref_all, nei_all)
    neq = pv.NormalEq(ctx, F, ui, uj)
    d = torch.device("cuda", 0)
    d_aa = torch.from_numpy(aa).to(d); d_t = torch.from_numpy(t).to(d)
    packed = torch.zeros(neq.size, dtype=torch.float64, device=d)
    torch.cuda.synchronize()
    ctx.set_poses_dev(F, d_aa.data_ptr(), d_t.data_ptr())
    neq.accumulate_dev(rs, packed.data_ptr(), pv.LOSS_HUBER, LOSS_A, zero_first=True)
    ctx.synchronize()
    out = packed.cpu()
    n = rs.n
    neq.close(); rs.close(); ctx.close()
    return out, n


def _worker(rank, world, port, q, F, cols):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf, n = _rank_packed(rank, world, F, cols)
    cnt = torch.tensor([n], dtype=torch.int64)
    dist.all_reduce(buf)
    dist.all_reduce(cnt)
    dist.barrier()
    if rank == 0:
        q.put((buf.numpy(), int(cnt.item())))
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_the_single_rank_normal_equations():
    import torch.multiprocessing as mp
    F, cols = 10, 256
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q, F, cols)) for r in range(2)]
    for p in procs:
        p.start()
    reduced, n2 = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single, n1 = _rank_packed(0, 1, F, cols)
    single = single.numpy()
    assert n1 == n2 and n1 > 1000
    assert reduced.shape == single.shape
    assert np.allclose(reduced, single, rtol=1e-12, atol=1e-12 * np.abs(single).max())


# ---- two ranks, two GPUs: RCCL through the library's own communicator (pvlm_comm_*) -----------------------------------------
def _rccl_worker(rank, world, port, q, F, cols, id_file):
    sys.path.insert(0, ROOT)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import time
    import torch
    import panovlm_amd as pv
    torch.cuda.set_device(rank)
    ctx = pv.Context(rank)
    if rank == 0:                                         # the 128-byte id travels through a file, as a launcher would carry it
        with open(id_file + ".tmp", "wb") as f:
            f.write(ctx.comm_unique_id())
        os.replace(id_file + ".tmp", id_file)
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 120:
            raise RuntimeError("no RCCL id from rank 0")
        time.sleep(0.05)
    comm = pv.Comm(ctx, world, rank, open(id_file, "rb").read())
    d = torch.device("cuda", rank)
    buf = torch.arange(1000, dtype=torch.float64, device=d) * (rank + 1)
    torch.cuda.synchronize()
    comm.allreduce_sum_f64(buf.data_ptr(), buf.numel())
    ctx.synchronize()
    q.put((rank, buf.cpu().numpy()))
    comm.close(); ctx.close()


def test_two_rccl_ranks_through_pvlm_comm():
    """pvlm_comm_create / pvlm_allreduce_sum_f64 with one process per GPU over RCCL: needs two GPUs (RCCL refuses two ranks on one
    device) — on a one-GPU box this test SKIPS, loudly: the path below has then still never run."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("NOT RUN: pvlm_comm_* over RCCL needs >= 2 GPUs, this box shows %d — the N > 1 RCCL path stays unexecuted here" % torch.cuda.device_count())
    import tempfile
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    with tempfile.TemporaryDirectory() as d:
        procs = [mpc.Process(target=_rccl_worker, args=(r, 2, 0, q, 0, 0, os.path.join(d, "rccl_id"))) for r in range(2)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=300) for _ in range(2))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    want = np.arange(1000, dtype=np.float64) * 3
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


def test_bench_two_ranks_on_one_gpu_runs_the_multi_rank_branch_of_bench_py():
    """Insurance for the day a multi-GPU node runs `bench.py --gpus N`: the N > 1 branch of TODAY's
    bench.py — launcher, rendezvous, sharded association, the all-reduce inside the timed steps, the per-rank table, the image-space block —
    executed as two ranks sharing this GPU (gloo stands in for RCCL, which refuses two ranks on one device).  The JSON line must carry the
    multi-rank fields and the summed normal equations of the last step must equal the one-rank run's (1e-12: the order of the sums differs)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--scans", "64", "--cols", "512", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-mvs", "--no-projection"]

    def run(gpus, env_extra):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus)] + common, capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    one = run(1, {})
    two = run(2, {"PVLM_BENCH_SHARED_GPU": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["value"] > 0 and two["steps"] == 3 and two["warmup"] == 1
    st = two["step"]
    assert st["allreduce_us"] is not None and st["allreduce_us"] > 0 and st["allreduce_doubles"] > 0 and st["comm"] == "torch" and "gloo" in st["comm_backend"]
    assert len(st["per_rank"]) == 2 and sum(r["evals"] for r in st["per_rank"]) == two["config"]["residual_blocks"] == one["config"]["residual_blocks"]
    assert all(r["pairs"] > 0 and r["fused_kernel_ms"] > 0 for r in st["per_rank"])
    for k in ("packed_sum", "packed_l1"):
        assert abs(st[k] - one["step"][k]) <= 1e-12 * abs(one["step"][k]), (k, st[k], one["step"][k])
    assert abs(two["config"]["robust_cost"] - one["config"]["robust_cost"]) <= 1e-12 * one["config"]["robust_cost"]
    assert two["image_space_all_ranks"]["ranks_measured"] == 2
    assert two["roofline"]["kernel"] == "k_eval_fused" and two["scaling"] == "strong"


def test_sparse_association_blocks_are_made_dense():
    """Raw scans as targets at full resolution: nine queries in ten are rejected by the reference's collinearity test (here, at a quarter of the resolution, a short
    search radius does the rejecting), yet every query has a row reserved in the batch's column block
    (the ordered placement inside k_fit_pairs needs no sizing round trip).  A block that ends up less than half full is replaced by a dense copy (k_compact_block;
    from 64 MB on by default, forced here): the residual set keeps the memory it uses, and evaluates to the same bits."""
    code = (
        "import numpy as np, sys; sys.path.insert(0, %r)\n"
        "import panovlm_amd as pv\nfrom panovlm_amd import synthetic as sy\n"
        "F = 4; scans = {k: sy.make_scan(k, cols=1024) for k in range(F)}\n"
        "ref, nei = sy.pair_list(F, 2)\n"
        "ctx = pv.Context(0); dev = {k: pv.Scan(ctx, s) for k, s in scans.items()}\n"
        "m0 = ctx.mem_info()['in_use']\n"
        "rs = ctx.assoc_point2plane([dev[int(r)] for r in ref], [dev[int(n)] for n in nei], 0.05, 0.12, kind=pv.POINT2PLANE_ANGLE, flags=pv.FLAG_NORMALIZE_DISTANCE)\n"
        "ctx.synchronize(); held = ctx.mem_info()['in_use'] - m0\n"
        "off, r, n, rows = rs.download()\n"
        "aa = np.zeros((F, 3)); t = np.zeros((F, 3)); ctx.set_poses(aa, t)\n"
        "res, J = rs.eval(jac=True); blocks = rs.pair_blocks(pv.LOSS_HUBER, 0.03)\n"
        "queries = sum(len(scans[int(k)]['flat_xyz']) for k in nei)\n"
        "np.savez(sys.argv[1], off=off, rows=rows, res=res, J=J, blocks=blocks, held=held, queries=queries)\n" % ROOT)
    out = {}
    for tag, mb in (("dense", "0"), ("reserved", "1000000")):
        path = "/tmp/pvlm_compact_%s_%d.npz" % (tag, os.getpid())
        subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, PVLM_ASSOC_COMPACT_MIN_MB=mb), timeout=600)
        out[tag] = dict(np.load(path))
        os.remove(path)
    for k in ("off", "rows", "res", "J", "blocks"):
        assert np.array_equal(out["dense"][k], out["reserved"][k]), k
    n, q = int(out["dense"]["off"][-1]), int(out["dense"]["queries"])
    assert 0 < n < 0.45 * q, (n, q)                                               # most queries were rejected
    assert out["reserved"]["held"] >= q * 56 and out["dense"]["held"] <= out["reserved"]["held"] - 0.4 * q * 56, (out["dense"]["held"], out["reserved"]["held"], q)
