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
