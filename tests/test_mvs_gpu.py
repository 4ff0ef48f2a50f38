"""GPU parity of the panoramic MVS kernels (scoring pass pvlm_mvs_init_conf_map: MVS::InitPatchMap + InitConfMap,
mvs/MVS.cpp:586-680, :774-923; checkerboard PatchMatch sweep; depth fusion filters) against the CPU oracle, through the
C ABI.  BIT FOR BIT: the kernels add the NCC sums in the reference's sequential order (wave_seq_sum) and both sides use
the correctly rounded float exp / sin / cos / acos, so scores, validity decisions, tie-breaks between hypotheses and the
random perturbations drawn after them are identical."""
import os
import sys

import numpy as np
import pytest

from tests.test_mvs_cpu import mvs_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import panovlm_amd as pv
    c = pv.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("hw,step,rows,cols", [(3, 1, 180, 360), (5, 2, 96, 192), (11, 2, 96, 192)])
def test_conf_map_matches_oracle(ctx, oracle, hw, step, rows, cols):
    (gray, depth, normal), neis, Rn, tn = mvs_scene(oracle, rows, cols)
    rng = np.random.default_rng(7)
    depth = depth * rng.uniform(0.9, 1.1, size=depth.shape).astype(np.float32)
    depth[5:9, 7:30] = 0
    keep = np.full(depth.shape, 5.0, np.float32)
    co, do, no = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, hw, step, conf=keep)
    cg, dg, ng = ctx.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, hw, step, conf=keep)
    assert np.array_equal(co == -1, cg == -1) and np.array_equal(co == 5.0, cg == 5.0)
    assert np.array_equal(dg, do) and np.array_equal(ng, no)
    valid = (co > -1) & (co != 5.0)
    assert valid.mean() > 0.6
    assert np.array_equal(cg, co), (int((cg != co).sum()), float(np.abs(cg[valid] - co[valid]).max()))      # bit for bit


def test_conf_map_with_geometric_consistency_matches_oracle(ctx, oracle):
    (gray, depth, normal), neis, Rn, tn, nd = mvs_scene(oracle, 120, 240, with_depths=True)
    rng = np.random.default_rng(9)
    depth = depth * rng.uniform(0.97, 1.03, size=depth.shape).astype(np.float32)
    co, do, no = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, 3, 1, nei_depths=nd)
    cg, dg, ng = ctx.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, 3, 1, nei_depths=nd)
    assert np.array_equal(co == -1, cg == -1) and np.array_equal(dg, do) and np.array_equal(ng, no)
    valid = co > -1
    # the depth-validity functor |depth0 - d| / depth0 < 0.03 and min(angle, 2) are decided in float on both sides: a
    # corner flipping across the 3 % threshold would show as a jump of up to 0.4 — none may occur
    assert np.array_equal(cg, co), (int((cg != co).sum()), float(np.abs(cg[valid] - co[valid]).max()))
    pho, _, _ = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, 3, 1)
    assert (pho[valid] - co[valid]).max() > 0.3          # the term is active in this scene


def test_conf_map_edge_cases(ctx, oracle):
    import panovlm_amd as pv
    (gray, depth, normal), neis, Rn, tn = mvs_scene(oracle, 64, 128)
    # no neighbours: every pixel with depth gets -1 and loses its hypothesis
    c, d, n = ctx.mvs_init_conf_map(gray, [], np.zeros((0, 9)), np.zeros((0, 3)), depth, normal)
    assert np.all(c == -1) and np.all(d == 0) and np.all(n == 0)
    # a flat (texture-less) reference image: sq0 is a sum of rounding residues, zero or ~1e-10 depending on the window;
    # InitConfMap tests `sq0 > 0` only (mvs/MVS.cpp:602), so the outcome is decided by the order of the float sums —
    # the reference's sequential order on both sides: identical maps
    flat = np.full_like(gray, 100)
    c, d, n = ctx.mvs_init_conf_map(flat, neis, Rn, tn, depth, normal)
    co, do, no = oracle.mvs_init_conf_map(flat, neis, Rn, tn, depth, normal)
    assert np.array_equal(c, co) and np.array_equal(d, do) and np.array_equal(n, no)
    with pytest.raises(pv.PvlmError):
        ctx.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, half_window=20, step=1)      # 41 x 41 texels > 256


def test_depth_filter_matches_oracle(ctx, oracle):
    """pvlm_mvs_filter_depth (FilterDepthImage + ProjectDepthConfToRef): bit-exact — the forward splat keeps the minimum
    range per pixel (order-free, atomicMin on the float bit pattern) and the votes are float comparisons."""
    from tests.test_mvs_cpu import _filter_scene
    for rows, cols in ((96, 192), (180, 360)):
        nd, Rn, tn, depth, conf, const = _filter_scene(oracle, rows, cols)
        for kw in (dict(conf=conf, depth_constant=const, thr=0.01), dict(thr=0.03)):
            do, co = oracle.mvs_filter_depth(nd, Rn, tn, depth, **kw)
            dg, cg = ctx.mvs_filter_depth(nd, Rn, tn, depth, **kw)
            assert np.array_equal(dg, do) and np.array_equal(cg, co)
            assert (do > 0).mean() > 0.1
    # no neighbours: nothing survives
    d0, _ = ctx.mvs_filter_depth([], np.zeros((0, 9)), np.zeros((0, 3)), depth)
    assert np.all(d0 == 0)


def test_depth_filter_refine_matches_oracle(ctx, oracle):
    """pvlm_mvs_filter_depth_refine (FilterDepthImageRefine + ProjectDepthConfToRef with confidences): bit-exact — the keyed
    64-bit atomicMin reproduces the sequential "last writer among the closest", the per-pixel fusion is the reference's
    float arithmetic (compiled with -ffp-contract=off)."""
    from tests.test_mvs_cpu import _refine_scene
    for rows, cols in ((96, 192), (180, 360)):
        nd, nc, Rn, tn, depth, conf, const = _refine_scene(oracle, rows, cols)
        for kw in (dict(depth_constant=const, thr=0.01), dict(thr=0.03, max_depth=float(np.median(depth[depth > 0])))):
            do, co, ca = oracle.mvs_filter_depth_refine(nd, nc, Rn, tn, depth, conf, **kw)
            dg, cg, cb = ctx.mvs_filter_depth_refine(nd, nc, Rn, tn, depth, conf, **kw)
            assert np.array_equal(do, dg) and np.array_equal(co, cg) and np.array_equal(ca, cb)
            assert 0.1 < (do > 0).mean() < 0.95
    # no neighbours: nothing can be confirmed; only depth_constant pixels come through, with confidence 1
    d0, c0, _ = ctx.mvs_filter_depth_refine([], [], np.zeros((0, 9)), np.zeros((0, 3)), depth, conf, depth_constant=const)
    keep = (const == 1) & (depth > 0)
    assert np.array_equal(d0[keep], depth[keep]) and np.all(c0[keep] == 1) and not d0[~keep].any()


def sweep_agreement(got, want, valid):
    """Fraction of valid pixels on which two sweep results agree (depth 1e-4 relative, normal and conf 1e-4 absolute)."""
    dg, ng, cg = got; do, no, co = want
    same = (np.abs(dg - do) <= 1e-4 * np.maximum(np.abs(do), 1e-3)) & (np.abs(ng - no).max(axis=2) <= 1e-4) & (np.abs(cg - co) <= 1e-4)
    return float(same[valid].mean())


def test_patchmatch_sweep_matches_oracle(ctx, oracle):
    """pvlm_mvs_propagate (k_mvs_propagate: one wave per pixel, process_pixel + wave-level ScorePixel) against the oracle:
    depth, normal and confidence maps identical (round 1: 99 % of the pixels to 1e-4 — tree-order sums and the device's
    float library functions made near-ties fall the other way)."""
    from tests.test_mvs_cpu import sweep_scene
    S = sweep_scene(oracle)
    args = (S["gray"], S["neis"], S["Rn"], S["tn"], S["depth"], S["normal"], S["conf"])
    valid = S["conf"] > -1
    err = lambda d: float(np.median(np.abs(d[valid] / S["truth"][valid] - 1)))
    for kw in (dict(max_iter=1, seed=5), dict(max_iter=1, seed=9, nei_depths=S["nd"], depth_constant=S["const"], conf_threshold=0.9)):
        want = oracle.mvs_propagate(*args, **kw)
        got = ctx.mvs_propagate(*args, **kw)
        agree = sweep_agreement(got, want, valid)
        diff = [int((a != b).sum()) for a, b in zip(got, want)]
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), (agree, diff)          # depth, normal, conf: bit for bit
        if "conf_threshold" not in kw:
            assert np.all(got[2][valid] >= S["conf"][valid])                                       # monotone, as on the CPU
        m = (S["const"] == 1) & valid
        if "depth_constant" in kw:
            assert np.array_equal(got[0][m], S["depth"][m])
        again = ctx.mvs_propagate(*args, **kw)
        assert all(np.array_equal(a, b) for a, b in zip(got, again))                              # same arguments, same maps
    d3 = ctx.mvs_propagate(*args, max_iter=3, seed=5)[0]
    assert err(d3) < 0.3 * err(S["depth"])


def test_sequential_sweep_matches_oracle(ctx, oracle):
    """pvlm_mvs_propagate_sequential — Propagate::SEQUENTIAL, the strategy config/Room.txt:90 and config/Floor.txt:88 select
    (MVS::PropagateSequential, mvs/MVS.cpp:1057-1097): the oracle walks the image pixel by pixel in raster order (back again on odd
    iterations), the GPU runs one launch per anti-diagonal.  Depth, normal and confidence maps identical, photometric and with the
    geometric term / depth_constant / threshold; also through the resident view set; and it is not the checkerboard sweep."""
    from panovlm_amd.api import MvsViews
    from tests.test_mvs_cpu import sweep_scene
    S = sweep_scene(oracle)
    args = (S["gray"], S["neis"], S["Rn"], S["tn"], S["depth"], S["normal"], S["conf"])
    valid = S["conf"] > -1
    err = lambda d: float(np.median(np.abs(d[valid] / S["truth"][valid] - 1)))
    for kw in (dict(max_iter=3, seed=5), dict(max_iter=2, seed=9, nei_depths=S["nd"], depth_constant=S["const"], conf_threshold=0.9)):
        want = oracle.mvs_propagate(*args, sequential=True, **kw)
        got = ctx.mvs_propagate(*args, sequential=True, **kw)
        diff = [int((a != b).sum()) for a, b in zip(got, want)]
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), diff
        assert all(np.array_equal(a, b) for a, b in zip(got, ctx.mvs_propagate(*args, sequential=True, **kw)))
        board = ctx.mvs_propagate(*args, **kw)
        assert not np.array_equal(board[0], got[0])
    swept = ctx.mvs_propagate(*args, sequential=True, max_iter=3, seed=5)
    assert err(swept[0]) < 0.3 * err(S["depth"])
    # resident views: the same launches on maps that stay in HBM
    rows, cols = S["depth"].shape
    V = MvsViews(ctx, rows, cols, 4)
    nei = [0, 2, 3]; ref = 1
    for k, b in enumerate(nei):
        V.upload(b, gray=S["neis"][k], depth=S["nd"][k], normal=np.zeros((rows, cols, 3), np.float32), conf=np.zeros((rows, cols), np.float32))
        V.snapshot_depth(b)
    V.upload(ref, gray=S["gray"], depth=S["depth"], normal=S["normal"], conf=S["conf"])
    V.estimate(ref, nei, S["Rn"], S["tn"], max_iter=3, seed=5, sequential=True)
    got = V.download(ref, ("depth", "normal", "conf"))
    assert np.array_equal(got["depth"], swept[0]) and np.array_equal(got["normal"], swept[1]) and np.array_equal(got["conf"], swept[2])
    # several views per launch (pvlm_mvs_views_estimate_sequential_batch: upstream runs this strategy with one image per thread):
    # two jobs with different references, neighbour lists, seeds and (for one) depth_constant == the two single calls
    import panovlm_amd as pv
    jobs = [dict(ref=1, nei=nei, R_nr=S["Rn"], t_nr=S["tn"], seed=5, depth_constant=S["const"]),
            dict(ref=2, nei=[0, 3], R_nr=S["Rn"][:2], t_nr=S["tn"][:2], seed=11)]
    init = {1: (S["depth"], S["normal"], S["conf"]), 2: (S["nd"][1], S["normal"], S["conf"])}
    single = {}
    for j in jobs:
        d0, n0, c0 = init[j["ref"]]
        V.upload(j["ref"], gray=S["gray"] if j["ref"] == 1 else S["neis"][1], depth=d0, normal=n0, conf=c0)
    for geo in (False, True):
        for j in jobs:
            V.upload(j["ref"], depth=init[j["ref"]][0], normal=init[j["ref"]][1], conf=init[j["ref"]][2])
        for j in jobs:
            V.estimate(j["ref"], j["nei"], j["R_nr"], j["t_nr"], max_iter=2, seed=j["seed"], sequential=True, use_geometry=geo, depth_constant=j.get("depth_constant"),
                       conf_threshold=0.5)
            single[j["ref"]] = V.download(j["ref"], ("depth", "normal", "conf"))
        for j in jobs:
            V.upload(j["ref"], depth=init[j["ref"]][0], normal=init[j["ref"]][1], conf=init[j["ref"]][2])
        V.estimate_sequential_batch(jobs, max_iter=2, use_geometry=geo, conf_threshold=0.5)
        for j in jobs:
            got = V.download(j["ref"], ("depth", "normal", "conf"))
            for k in ("depth", "normal", "conf"):
                assert np.array_equal(got[k], single[j["ref"]][k]), (geo, j["ref"], k)
        assert not np.array_equal(single[1]["depth"], init[1][0]) and not np.array_equal(single[2]["depth"], init[2][0])
    with pytest.raises(pv.PvlmError):
        V.estimate_sequential_batch([jobs[0], jobs[0]])            # the same reference twice in one batch
    V.close()


def test_sequential_batch_with_six_neighbours_matches_oracle(ctx, oracle):
    """More neighbour images than the four a quad of threads takes in one round (QuadScorer visits them in groups of four, the last
    group partly filled) and than the float4 strips of the wave form hold: two jobs of the batched sequential sweep against the
    oracle's raster walk.  Re-run with the four-threads-per-pixel form forced by test_other_launch_forms_give_the_same_maps."""
    from panovlm_amd.api import MvsViews
    rows, cols = 64, 128
    (gray, depth, normal), neis, Rn, tn = mvs_scene(oracle, rows, cols, n_views=7)
    rng = np.random.default_rng(3)
    d0 = (depth * rng.uniform(0.92, 1.08, size=depth.shape)).astype(np.float32)
    c0, d1, n1 = oracle.mvs_init_conf_map(gray, neis, Rn, tn, d0, normal, 3, 1)
    want = [oracle.mvs_propagate(gray, neis, Rn, tn, d1, n1, c0, max_iter=2, seed=sd, sequential=True, conf_threshold=0.3) for sd in (9, 10)]
    V = MvsViews(ctx, rows, cols, len(neis) + 2)
    zero3 = np.zeros((rows, cols, 3), np.float32); zero1 = np.zeros((rows, cols), np.float32)
    for k, g in enumerate(neis):
        V.upload(k, gray=g, depth=zero1, normal=zero3, conf=zero1)
    refs = [len(neis), len(neis) + 1]
    for r in refs:
        V.upload(r, gray=gray, depth=d1, normal=n1, conf=c0)
    jobs = [dict(ref=r, nei=list(range(len(neis))), R_nr=Rn, t_nr=tn, seed=sd) for r, sd in zip(refs, (9, 10))]
    V.estimate_sequential_batch(jobs, max_iter=2, conf_threshold=0.3)
    for r, w in zip(refs, want):
        got = V.download(r, ("depth", "normal", "conf"))
        assert np.array_equal(got["depth"], w[0]) and np.array_equal(got["normal"], w[1]) and np.array_equal(got["conf"], w[2])
    assert not np.array_equal(want[0][0], want[1][0])
    V.close()


def test_other_launch_forms_give_the_same_maps():
    """The image-space kernels have several launch forms chosen by size: one pixel per thread (scoring pass, colour pass) or one wave
    per pixel, and for a batched anti-diagonal one wave per pixel or four threads per pixel from PVLM_MVS_QUAD_MIN pixels per
    diagonal.  The switches are read once per process, so the oracle comparisons of this file are re-run in child processes with
    the non-default forms forced: every form must give the oracle's maps bit for bit."""
    import subprocess
    here = os.path.abspath(__file__)
    for env in ({"PVLM_MVS_QUAD_MIN": "1"}, {"PVLM_MVS_LANE": "0"}, {"PVLM_MVS_LANE_TABLE_MB": "4"}, {"PVLM_MVS_FLOW": "0"}, {"PVLM_MVS_FLOW": "1"}):   # FLOW 0: one launch per anti-diagonal, 1: data-flow launch, a wave per pixel (default at these sizes: 2)
        r = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k", "sweep_matches_oracle or conf_map_matches_oracle or sequential", "-p", "no:cacheprovider"],
                           env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, (env, r.stdout[-1500:])


def test_data_flow_sweep_on_odd_shapes():
    """The persistent data-flow launches of the sequential sweep (K13p / K13q: ticket order over the anti-diagonals, per-pixel hand-off
    cells) against one launch per anti-diagonal on shapes the oracle scenes do not have: taller than wide, odd sizes, barely larger than
    the window, and three iterations (forward, backward, forward).  The form is read once per process: child processes, one per form;
    the maps must agree bit for bit (sha256 of depth | normal | conf)."""
    import json, subprocess
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mvs_seq_bench.py")
    for rows, cols in ((40, 24), (17, 33), (9, 64), (64, 201)):
        sums = {}
        for form in ("0", "1", "2"):
            r = subprocess.run([sys.executable, tool, "--rows", str(rows), "--cols", str(cols), "--iters", "3", "--neighbors", "2"],
                               env=dict(os.environ, PVLM_MVS_FLOW=form), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            assert r.returncode == 0, (rows, cols, form, r.stderr[-800:])
            sums[form] = json.loads(r.stdout.strip().splitlines()[-1])["sha256"]
        assert sums["0"] == sums["1"] == sums["2"], (rows, cols, sums)


def test_resident_views_equal_the_per_call_entry_points(ctx, oracle):
    """pvlm_mvs_views_*: the same kernels on maps that stay in HBM — every stage must equal the per-call API bit for bit
    (scoring pass, sweep incl. geometric consistency / depth_constant / threshold, fusion filter with its in-place conf)."""
    import time
    from panovlm_amd.api import MvsViews
    from tests.test_mvs_cpu import sweep_scene
    S = sweep_scene(oracle)
    rows, cols = S["depth"].shape
    nei = [0, 2, 3]; ref = 1                                       # view ids of the scene: the reference is view 1
    V = MvsViews(ctx, rows, cols, 4)
    rng = np.random.default_rng(8)
    nconf = [rng.uniform(0, 1, size=(rows, cols)).astype(np.float32) for _ in nei]
    nnormal = np.zeros((rows, cols, 3), np.float32)
    for k, b in enumerate(nei):                                    # neighbours: grey image, their own depth map and confidence
        V.upload(b, gray=S["neis"][k], depth=S["nd"][k], normal=nnormal, conf=nconf[k])
        V.snapshot_depth(b)                                        # depth_filter <- depth: what use_geometry reads
    # ---- scoring pass (InitConfMap), photometric then geometric
    d0 = (S["truth"] * np.random.default_rng(7).uniform(0.9, 1.1, size=S["truth"].shape)).astype(np.float32)
    for geo in (False, True):
        V.upload(ref, gray=S["gray"], depth=d0, normal=S["normal"], conf=np.zeros((rows, cols), np.float32))
        V.estimate(ref, nei, S["Rn"], S["tn"], use_geometry=geo, max_iter=-1)
        got = V.download(ref, ("depth", "normal", "conf"))
        c, d, n = ctx.mvs_init_conf_map(S["gray"], S["neis"], S["Rn"], S["tn"], d0, S["normal"], 3, 1, nei_depths=S["nd"] if geo else None)
        assert np.array_equal(got["conf"], c) and np.array_equal(got["depth"], d) and np.array_equal(got["normal"], n)
    # ---- sweep
    for kw in (dict(max_iter=2, seed=5), dict(max_iter=1, seed=9, use_geometry=True, depth_constant=S["const"], conf_threshold=0.9)):
        V.upload(ref, depth=S["depth"], normal=S["normal"], conf=S["conf"])
        V.estimate(ref, nei, S["Rn"], S["tn"], **kw)
        got = V.download(ref, ("depth", "normal", "conf"))
        pk = dict(kw); geo = pk.pop("use_geometry", False)
        want = ctx.mvs_propagate(S["gray"], S["neis"], S["Rn"], S["tn"], S["depth"], S["normal"], S["conf"], nei_depths=S["nd"] if geo else None, **pk)
        assert np.array_equal(got["depth"], want[0]) and np.array_equal(got["normal"], want[1]) and np.array_equal(got["conf"], want[2])
    # ---- fusion filter on the swept state (still resident), with and without depth_constant
    swept = V.download(ref, ("depth", "conf"))
    cref = np.clip(swept["conf"], 0, None)
    for kw in (dict(thr=0.02), dict(thr=0.01, depth_constant=S["const"], max_depth=float(np.median(swept["depth"][swept["depth"] > 0])))):
        V.upload(ref, conf=cref)
        V.filter_refine(ref, nei, S["Rn"], S["tn"], **kw)
        got = V.download(ref, ("conf", "depth_filter", "conf_filter"))
        dw, cw, ca = ctx.mvs_filter_depth_refine(S["nd"], nconf, S["Rn"], S["tn"], swept["depth"], cref, **kw)
        assert np.array_equal(got["depth_filter"], dw) and np.array_equal(got["conf_filter"], cw) and np.array_equal(got["conf"], ca)
        assert 0.02 < (dw > 0).mean() < 0.98
    # ---- bad arguments are refused, not executed
    import panovlm_amd as pv
    for bad in (dict(ref=1, nei=[1, 2]), dict(ref=4, nei=[0]), dict(ref=0, nei=[7])):
        with pytest.raises(pv.PvlmError):
            V.estimate(bad["ref"], bad["nei"], S["Rn"][:len(bad["nei"])], S["tn"][:len(bad["nei"])])
    # what residency buys at this size: wall time of one sweep iteration, maps in HBM vs the per-call entry point
    t0 = time.perf_counter(); V.estimate(ref, nei, S["Rn"], S["tn"], max_iter=1, seed=3); V.download(ref, ("conf",)); t_res = time.perf_counter() - t0
    t0 = time.perf_counter(); ctx.mvs_propagate(S["gray"], S["neis"], S["Rn"], S["tn"], S["depth"], S["normal"], S["conf"], max_iter=1, seed=3); t_call = time.perf_counter() - t0
    print("resident %.2f ms, per call %.2f ms" % (t_res * 1e3, t_call * 1e3))
    V.close()



def test_init_depth_normal_and_remove_small_segments_match_oracle(ctx, oracle):
    """MVS::InitDepthNormal with the LiDAR depth prior (mvs/MVS.cpp:496-584; BASELINE config 5 "LiDAR-seeded depth priors") and
    MVS::RemoveSmallSegments (:1504-1577): bit for bit against the oracle, then the pieces in the order the reference's driver
    chains them (:436-470, :93-140): LiDAR prior -> InitDepthNormal -> InitConfMap -> sweep -> RemoveSmallSegments."""
    from panovlm_amd import synthetic as sy
    from tests.test_mvs_cpu import mvs_scene
    rows, cols = 180, 360
    cloud = np.concatenate([sy.make_scan(k, cols=512)["local_xyz"] for k in (0, 1)])
    lidar16 = ctx.project_lidar_depth(rows, cols, cloud, np.eye(4), 2)
    assert np.array_equal(lidar16, oracle.project_lidar_depth(rows, cols, cloud, np.eye(4), 2)) and 0.05 < (lidar16 > 0).mean() < 0.9
    rng = np.random.default_rng(4)
    mask = (rng.uniform(size=(rows, cols)) > 0.03).astype(np.float32)
    for kw in (dict(lidar_depth16=lidar16, mask=mask, keep_lidar_constant=True, seed=7), dict(lidar_depth16=lidar16, keep_lidar_constant=False, seed=8),
               dict(mask=mask, seed=9, min_depth=0.5, max_depth=12.0)):
        dg, ng, cg = ctx.mvs_init_depth_normal(rows, cols, **kw)
        do, no, co = oracle.mvs_init_depth_normal(rows, cols, **kw)
        assert np.array_equal(dg, do) and np.array_equal(ng, no) and np.array_equal(cg, co)
        kept = np.ones((rows, cols), bool) if kw.get("mask") is None else mask >= 1
        assert np.all(dg[~kept] == 0) and np.all(ng[~kept] == 0)
        assert np.allclose(np.linalg.norm(ng[kept], axis=1), 1, atol=1e-5) and dg[kept].min() >= kw.get("min_depth", 0.1) - 1e-6
        if kw.get("lidar_depth16") is not None:
            seeded = (lidar16 > 0) & kept
            assert np.array_equal(dg[seeded], (lidar16[seeded].astype(np.float32) / np.float32(256)))
            assert np.array_equal(cg == 1, lidar16 > 0) if kw["keep_lidar_constant"] else not cg.any()
    # RemoveSmallSegments on a depth map with islands, holes and a smooth ramp
    (gray, depth, normal), neis, Rn, tn = mvs_scene(oracle, rows, cols)
    d = depth.copy()
    d[rng.uniform(size=d.shape) < 0.08] = 0                        # holes
    d[40:44, 50:58] *= 1.5; d[100:103, 200:204] *= 0.6             # islands smaller than min_segment at another depth
    conf = rng.uniform(0, 1, size=d.shape).astype(np.float32)
    for thr, mseg in ((0.01, 100), (0.05, 20), (0.002, 400)):
        a = ctx.mvs_remove_small_segments(d, normal, conf, thr, mseg)
        b = oracle.mvs_remove_small_segments(d, normal, conf, thr, mseg)
        assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3] > 0
        assert np.all(a[2][a[0] == 0] == -1) and np.all(a[1][a[0] == 0] == 0)
    assert np.all(ctx.mvs_remove_small_segments(d, normal, conf, 0.01, 100)[0][40:44, 50:58] == 0)
    # the driver's chain on the LiDAR-seeded state
    d0, n0, c0 = ctx.mvs_init_depth_normal(rows, cols, lidar_depth16=lidar16, seed=3, max_depth=8.0)
    cf, d1, n1 = ctx.mvs_init_conf_map(gray, neis, Rn, tn, d0, n0, 3, 1)
    got = ctx.mvs_propagate(gray, neis, Rn, tn, d1, n1, cf, depth_constant=c0, max_iter=2, seed=3, max_depth=8.0)
    want = oracle.mvs_propagate(gray, neis, Rn, tn, *oracle.mvs_init_conf_map(gray, neis, Rn, tn, d0, n0, 3, 1)[1:], oracle.mvs_init_conf_map(gray, neis, Rn, tn, d0, n0, 3, 1)[0],
                                depth_constant=c0, max_iter=2, seed=3, max_depth=8.0)
    assert all(np.array_equal(x, y) for x, y in zip(got, want))
    fin = ctx.mvs_remove_small_segments(got[0], got[1], got[2], 0.01, 100)
    assert np.array_equal(fin[0], oracle.mvs_remove_small_segments(want[0], want[1], want[2], 0.01, 100)[0])


@pytest.mark.parametrize("rows,cols", [(60, 120), (37, 101), (720, 1440)])
def test_depth_to_cloud_matches_oracle(ctx, oracle, rows, cols):
    """pvlm_mvs_depth_to_cloud (MVS::DepthImageToCloud / DepthNormalToCloud, mvs/MVS.cpp:2073-2142): same points, colours, normals, in
    the same (raster) order, bit for bit — 37 x 101 leaves a ragged last workgroup, 720 x 1440 takes four rounds of the block scan."""
    from tests import synth
    from panovlm_amd.api import MvsViews
    depth, bgr, normal, T = synth.cloud_scene(np.random.default_rng(rows), rows, cols, 20.0)
    want = oracle.mvs_depth_to_cloud(depth, bgr, T, 20.0)
    got = ctx.mvs_depth_to_cloud(depth, bgr, T, 20.0)
    assert len(want[0]) > 0.2 * rows * cols
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    wantn = oracle.mvs_depth_to_cloud(depth, bgr, T, 20.0, filter_sky=False, normal=normal)
    gotn = ctx.mvs_depth_to_cloud(depth, bgr, T, 20.0, filter_sky=False, normal=normal)
    assert len(wantn[0]) > len(want[0])
    assert all(np.array_equal(a, b) for a, b in zip(gotn, wantn))
    # 3 x 4 pose, a different max_depth
    w2 = oracle.mvs_depth_to_cloud(depth, bgr, T[:3], 7.5); g2 = ctx.mvs_depth_to_cloud(depth, bgr, T[:3], 7.5)
    assert 0 < len(w2[0]) < len(want[0]) and np.array_equal(g2[0], w2[0]) and np.array_equal(g2[1], w2[1])
    # the resident view: depth_filter (uploaded as depth, then snapshot) and depth, unit rays from the view set's table
    V = MvsViews(ctx, rows, cols, 2)
    V.upload(1, gray=np.zeros((rows, cols), np.uint8), depth=depth, normal=normal, conf=np.zeros((rows, cols), np.float32))
    gv = V.depth_to_cloud(1, bgr, T, 20.0, filter_sky=False, use_filtered_depth=False, with_normal=True)
    assert all(np.array_equal(a, b) for a, b in zip(gv, wantn))
    V.snapshot_depth(1)
    gv = V.depth_to_cloud(1, bgr, T, 20.0)
    assert np.array_equal(gv[0], want[0]) and np.array_equal(gv[1], want[1])
    V.close()


def test_depth_to_cloud_edge_cases(ctx, oracle):
    rows, cols = 16, 48
    bgr = np.full((rows, cols, 3), 90, np.uint8); T = np.eye(4)
    none = ctx.mvs_depth_to_cloud(np.zeros((rows, cols), np.float32), bgr, T)
    assert none[0].shape == (0, 3) and none[1].shape == (0, 3)
    every = ctx.mvs_depth_to_cloud(np.full((rows, cols), 2.0, np.float32), bgr, T)
    want = oracle.mvs_depth_to_cloud(np.full((rows, cols), 2.0, np.float32), bgr, T)
    assert len(every[0]) == rows * cols and np.array_equal(every[0], want[0])
    assert np.allclose(np.linalg.norm(every[0], axis=1), 2.0, atol=1e-5)
    one = np.zeros((1, 1), np.float32) + 3
    g = ctx.mvs_depth_to_cloud(one, np.zeros((1, 1, 3), np.uint8), T); w = oracle.mvs_depth_to_cloud(one, np.zeros((1, 1, 3), np.uint8), T)
    assert len(g[0]) == 1 and np.array_equal(g[0], w[0])
    with pytest.raises(Exception):
        ctx.mvs_depth_to_cloud(np.zeros((0, 4), np.float32), np.zeros((0, 4, 3), np.uint8), T)
