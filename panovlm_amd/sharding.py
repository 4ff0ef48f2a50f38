"""Frame sharding of the scan-pair list across ranks.

The unit of independent work is one ordered scan pair (ref i, nei n).  Pairs are block-partitioned
by REFERENCE scan so that the association of a rank only needs its own reference scans plus their
neighbours; the only exchange is one all-reduce(sum) of the packed normal-equation buffer
[Hdiag F x 36 | Hoff U x 36 | g F x 6 | cost] per LM iteration (include/pvlm.h, pvlm_neq_*).
numpy only — used by bench.py on the GPU and by the gloo tests on the CPU."""
import numpy as np


def shard_range(F, rank, world):
    return (F * rank) // world, (F * (rank + 1)) // world


def shard_pairs(ref, nei, F, rank, world):
    """Pairs whose reference scan lies in this rank's block, order preserved."""
    lo, hi = shard_range(F, rank, world)
    m = (ref >= lo) & (ref < hi)
    return ref[m], nei[m]


def unordered_pairs(ref, nei):
    """Sorted unique (i < j) pose pairs: the off-diagonal block structure shared by every rank."""
    a = np.minimum(ref, nei).astype(np.int64); b = np.maximum(ref, nei).astype(np.int64)
    key = np.unique(a * (int(b.max()) + 1 if len(b) else 1) + b)
    base = int(b.max()) + 1 if len(b) else 1
    return (key // base).astype(np.int32), (key % base).astype(np.int32)


def packed_size(F, U):
    return F * 36 + U * 36 + F * 6 + 1


def unpack(packed, F, U):
    Hd = packed[:F * 36].reshape(F, 6, 6); Ho = packed[F * 36:(F + U) * 36].reshape(U, 6, 6)
    g = packed[(F + U) * 36:(F + U) * 36 + F * 6].reshape(F, 6)
    return Hd, Ho, g, float(packed[-1])


def pack_from_pair_blocks(blocks, ref, nei, F, ui, uj):
    """Host reference of the device gather (k_neq_gather): pair blocks (P x 121) -> packed buffer."""
    U = len(ui)
    out = np.zeros(packed_size(F, U))
    Hd, Ho, g, _ = unpack(out, F, U)   # views
    Hd = out[:F * 36].reshape(F, 6, 6); Ho = out[F * 36:(F + U) * 36].reshape(U, 6, 6); g = out[(F + U) * 36:-1].reshape(F, 6)
    lut = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(ui, uj))}
    for p in range(len(ref)):
        r, n = int(ref[p]), int(nei[p])
        b = blocks[p]
        Hd[r] += b[0:36].reshape(6, 6); Hd[n] += b[72:108].reshape(6, 6)
        g[r] += b[108:114]; g[n] += b[114:120]
        Hrn = b[36:72].reshape(6, 6)
        if r < n:
            Ho[lut[(r, n)]] += Hrn
        else:
            Ho[lut[(n, r)]] += Hrn.T
        out[-1] += b[120]
    return out


# ---- per-view sharding of the MVS ("Config 5 (MVS) shards per reference view") ----
def shard_views(n_views, rank, world, cost=None):
    """Reference views of this rank.  Views are independent (mvs/MVS.cpp:93-117: one Initialize + EstimateDepthMapSingle
    per reference view; neighbour images are only read), so there is no exchange: every rank keeps the grey images it
    needs and writes its own depth maps.  Without `cost` a contiguous block; with per-view costs (e.g. pixels carrying a
    depth prior) the longest-processing-time greedy assignment, identical on every rank."""
    if cost is None:
        lo, hi = shard_range(n_views, rank, world)
        return list(range(lo, hi))
    order = sorted(range(n_views), key=lambda v: (-float(cost[v]), v))
    load = [0.0] * world; mine = []
    for v in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += float(cost[v])
        if r == rank:
            mine.append(v)
    return sorted(mine)


def estimate_depth_maps(worker, views, neighbors, rank=0, world=1, half_window=3, step=1, pho_iters=3, conf_threshold=-0.7, min_depth=0.1, max_depth=20.0,
                        seed=1, cost=None, sequential=False):
    """The photometric pass of MVS::EstimateDepthMaps (mvs/MVS.cpp:93-117) for this rank's views: InitConfMap, then
    EstimateDepthMapSingle(view, CHECKER_BOARD or — sequential=True, config propagate_strategy = 2 — SEQUENTIAL, pho_iters, conf_threshold, false).  worker: a panovlm_amd.Context (the two
    calls run on its GPU).  views[v] = dict(gray, depth, normal) with an initialised depth / normal hypothesis;
    neighbors[v] = list of (neighbour view, R_nr (3x3), t_nr (3)).  Returns {view: (depth, normal, conf)}."""
    out = {}
    for v in shard_views(len(views), rank, world, cost):
        nb = neighbors[v]
        grays = [views[n]["gray"] for n, _, _ in nb]
        R = np.array([r for _, r, _ in nb], np.float32).reshape(-1, 9); t = np.array([x for _, _, x in nb], np.float32).reshape(-1, 3)
        conf, depth, normal = worker.mvs_init_conf_map(views[v]["gray"], grays, R, t, views[v]["depth"], views[v]["normal"], half_window, step)
        out[v] = worker.mvs_propagate(views[v]["gray"], grays, R, t, depth, normal, conf, half_window=half_window, step=step, min_depth=min_depth,
                                      max_depth=max_depth, seed=seed + v, max_iter=pho_iters, conf_threshold=conf_threshold, sequential=sequential)
    return out



def merge_depth_images(worker, depths, bgrs, poses, skip=1, max_depth=20.0, rank=0, world=1):
    """MVS::MergeDepthImages(skip) (mvs/MVS.cpp:2144-2166; FuseDepthMaps :224-227 calls it with skip = 2): the clouds of every skip-th frame
    with a depth map (DepthImageToCloud, worker.mvs_depth_to_cloud) appended in frame order — upstream appends per OpenMP thread, i.e. in
    frame order with one thread.  depths[i]: rows x cols float32 or None (an empty depth map: skipped), bgrs[i]: rows x cols x 3 uint8,
    poses[i]: T_wc (3 x 4 or 4 x 4).  This rank takes a contiguous block of the selected frames: concatenating the ranks' results in rank
    order gives the single-rank cloud, no exchange.  Returns (xyz n x 3 float32, rgb n x 3 uint8)."""
    assert skip >= 1
    chosen = [i for i in range(0, len(depths), skip) if depths[i] is not None]
    lo, hi = shard_range(len(chosen), rank, world)
    xyz, rgb = [np.zeros((0, 3), np.float32)], [np.zeros((0, 3), np.uint8)]
    for i in chosen[lo:hi]:
        p, c = worker.mvs_depth_to_cloud(depths[i], bgrs[i], poses[i], max_depth)[:2]
        xyz.append(p); rgb.append(c)
    return np.concatenate(xyz), np.concatenate(rgb)
