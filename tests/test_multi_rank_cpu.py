"""N > 1 path on CPU: two gloo ranks shard the pair list by reference scan, build their packed
normal-equation buffers (oracle evaluation stands in for the HIP kernels, which need a GPU), all-reduce
them and must reproduce the single-rank buffer.  Also checks the sharding helpers bench.py uses."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _problem():
    from tests import synth
    from panovlm_amd import synthetic as sy
    rng = np.random.default_rng(42)
    F = 8
    aa, t = synth.random_poses(rng, F)
    ref, nei = sy.pair_list(F, 4)
    counts = rng.integers(5, 60, size=len(ref))
    rows, off = synth.random_resset(rng, 1, aa, t, ref, nei, counts)
    return F, aa, t, ref, nei, rows, off


def _packed_for(oracle, F, aa, t, ref, nei, rows, off, ui, uj):
    from panovlm_amd import sharding as sh
    from tests import synth
    rid, nid = synth.expand_ids(off, ref, nei)
    r, J = oracle.evaluate(1, synth.oracle_rows(1, rows), rid, nid, aa, t, normalize=True)
    blocks = synth.pair_blocks_from_jacobian(r, J, off, 1, 2 * np.pi / 180)
    return sh.pack_from_pair_blocks(blocks, ref, nei, F, ui, uj)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from panovlm_amd import sharding as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F, aa, t, ref, nei, rows, off = _problem()
    ui, uj = sh.unordered_pairs(ref, nei)
    lo, hi = sh.shard_range(F, rank, world)
    keep = np.flatnonzero((ref >= lo) & (ref < hi))
    my_ref, my_nei = sh.shard_pairs(ref, nei, F, rank, world)
    assert np.array_equal(my_ref, ref[keep])
    my_rows = np.concatenate([rows[off[p]:off[p + 1]] for p in keep]) if len(keep) else np.zeros((0, 7))
    my_off = np.concatenate([[0], np.cumsum([off[p + 1] - off[p] for p in keep])]).astype(np.int64)
    packed = _packed_for(orc, F, aa, t, my_ref, my_nei, my_rows, my_off, ui, uj)
    buf = torch.from_numpy(packed.copy())
    dist.all_reduce(buf)
    dist.barrier()
    if rank == 0:
        q.put(buf.numpy())
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_single_rank(oracle):
    import torch.multiprocessing as mp
    from panovlm_amd import sharding as sh
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    reduced = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    F, aa, t, ref, nei, rows, off = _problem()
    ui, uj = sh.unordered_pairs(ref, nei)
    full = _packed_for(oracle, F, aa, t, ref, nei, rows, off, ui, uj)
    assert reduced.shape == full.shape == (sh.packed_size(F, len(ui)),)
    assert np.allclose(reduced, full, rtol=1e-12, atol=1e-12 * np.abs(full).max())
    Hd, Ho, g, cost = sh.unpack(full, F, len(ui))
    assert cost > 0 and np.all(np.linalg.eigvalsh(Hd[3]) > -1e-9)


def test_sharding_covers_every_pair_once():
    from panovlm_amd import sharding as sh
    from panovlm_amd import synthetic as sy
    for F, nb, world in [(8, 4, 2), (37, 8, 4), (64, 8, 8), (5, 4, 8)]:
        ref, nei = sy.pair_list(F, nb)
        seen = []
        for r in range(world):
            a, b = sh.shard_pairs(ref, nei, F, r, world)
            seen += list(zip(a.tolist(), b.tolist()))
        assert sorted(seen) == sorted(zip(ref.tolist(), nei.tolist())) and len(set(seen)) == len(seen)
        ui, uj = sh.unordered_pairs(ref, nei)
        assert np.all(ui < uj) and len(set(zip(ui.tolist(), uj.tolist()))) == len(ui)
        assert set(zip(ui.tolist(), uj.tolist())) == {(min(a, b), max(a, b)) for a, b in zip(ref.tolist(), nei.tolist())}


# ---- MVS: per-view sharding, no exchange ----------------------------------------------------------------------
def _mvs_views(oracle, n=4, rows=48, cols=96):
    from tests import synth
    poses = [(synth.rodrigues(np.array([0.02 * k, 0.25 * k - 0.3, 0.01])), np.array([0.35 * k - 0.5, 0.04 * k, 0.25 * k - 0.3])) for k in range(n)]
    rng = np.random.default_rng(3)
    views = []
    for R, t in poses:
        g, d, nrm = synth.render_panorama(oracle, rows, cols, R, t)
        views.append(dict(gray=g, depth=(d * rng.uniform(0.93, 1.07, size=d.shape)).astype(np.float32), normal=nrm))
    nb = [[(k,) + synth.relative_pose(poses[v][0], poses[v][1], poses[k][0], poses[k][1]) for k in range(n) if k != v] for v in range(n)]
    return views, nb


class _OracleWorker:
    """Stands in for panovlm_amd.Context in the CPU test of the sharding logic (the GPU calls themselves are covered by
    tests/test_mvs_gpu.py): same two methods, backed by the oracle."""
    def __init__(self, oracle):
        self.o = oracle

    def mvs_init_conf_map(self, *a, **k):
        return self.o.mvs_init_conf_map(*a, **k)

    def mvs_propagate(self, *a, **k):
        return self.o.mvs_propagate(*a, **k)

    def mvs_depth_to_cloud(self, *a, **k):
        return self.o.mvs_depth_to_cloud(*a, **k)


def _cloud_frames(n=7, rows=24, cols=48):
    from tests import synth
    rng = np.random.default_rng(5)
    depths, bgrs, poses = [], [], []
    for k in range(n):
        d, c, _, T = synth.cloud_scene(rng, rows, cols)
        T[:3, 3] += k
        depths.append(None if k == 2 else d); bgrs.append(c); poses.append(T)      # frame 2 has no depth map: skipped, as upstream
    return depths, bgrs, poses


def _mvs_worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import oracle
    from panovlm_amd import sharding as sh
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    views, nb = _mvs_views(oracle)
    mine = sh.estimate_depth_maps(_OracleWorker(oracle), views, nb, rank, world, pho_iters=1, seed=11)
    cloud = sh.merge_depth_images(_OracleWorker(oracle), *_cloud_frames(), skip=2, rank=rank, world=world)
    gathered = [None] * world
    dist.all_gather_object(gathered, ({v: d[0] for v, d in mine.items()}, cloud))     # collecting the results is the only communication
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


def test_mvs_view_sharding_two_ranks(oracle):
    import torch.multiprocessing as mp
    from panovlm_amd import sharding as sh
    for n, world in [(4, 2), (7, 3), (3, 8)]:
        seen = sum((sh.shard_views(n, r, world) for r in range(world)), [])
        assert sorted(seen) == list(range(n))
        cost = [5, 1, 1, 1, 9, 2, 2][:n]
        parts = [sh.shard_views(n, r, world, cost) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(n))
        loads = [sum(cost[v] for v in p) for p in parts]
        assert max(loads) <= max(max(cost), -(-sum(cost) // world) + max(cost))          # LPT bound
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mvs_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    views, nb = _mvs_views(oracle)
    single = sh.estimate_depth_maps(_OracleWorker(oracle), views, nb, 0, 1, pho_iters=1, seed=11)
    # MergeDepthImages(2): frames 0, 4, 6 (2 has no depth map), rank 0 takes two of them; rank order = frame order
    whole = sh.merge_depth_images(_OracleWorker(oracle), *_cloud_frames(), skip=2)
    parts = [g[1] for g in gathered]
    assert all(len(p[0]) > 0 for p in parts)
    assert np.array_equal(np.concatenate([p[0] for p in parts]), whole[0]) and np.array_equal(np.concatenate([p[1] for p in parts]), whole[1])
    depths, bgrs, poses = _cloud_frames()
    by_hand = [oracle.mvs_depth_to_cloud(depths[i], bgrs[i], poses[i], 20.0) for i in (0, 4, 6)]
    assert np.array_equal(whole[0], np.concatenate([b[0] for b in by_hand]))
    gathered = [g[0] for g in gathered]
    merged = {}
    for part in gathered:
        assert not (set(part) & set(merged))
        merged.update(part)
    assert sorted(merged) == [0, 1, 2, 3] and sorted(gathered[0]) == [0, 1] and sorted(gathered[1]) == [2, 3]
    for v in range(4):
        assert np.array_equal(merged[v], single[v][0])                                    # a view's result does not depend on who computed it

