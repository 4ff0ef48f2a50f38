"""GPU parity of the residual/Jacobian kernels (K5 materialise, K6 fused) against the CPU oracle,
through the C ABI.  Tolerance: 1e-6 relative (BASELINE.json north_star), nothing added to it.

Arbiter = the reference's statements (Jet AutoDiff through its rotation chain, acos of the clamped cosine) evaluated in
x87 extended precision (oracle AutoDiffEvaluateExt): the exact value of upstream's formula to ~1e-19.  Upstream's own
double evaluation is the LESS accurate side near r -> 0 (acos near 1 loses eps / r; round 1 widened the gate by that
amount) and is itself held to the same 1e-6 here.  Decisions (the `dis < 1e-3 -> 0` early-out) are compared exactly,
and the blocks that sit ON a threshold are counted, not tolerated."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

RTOL = 1e-6


ATOL = 1e-13      # absolute floor for residuals that are exactly or nearly zero (radians / metres)


@pytest.fixture(scope="module")
def ctx():
    import panovlm_amd as pv
    c = pv.Context(0)
    yield c
    c.close()


CASES = [(0, False), (1, False), (1, True), (2, False), (3, False), (3, True), (4, False), (5, False)]


def _build(rng, kind, F=7, P=23, max_count=300):
    aa, t = synth.random_poses(rng, F)
    ref, nei = synth.random_pairs(rng, F, P)
    counts = rng.integers(0, max_count, size=P)
    counts[0] = 0          # empty segment
    counts[1] = 1          # odd, single-row segment
    counts[2] = 1025       # spans several 512-row iterations, odd tail
    rows, off = synth.random_resset(rng, kind, aa, t, ref, nei, counts)
    return aa, t, ref, nei, rows, off


@pytest.mark.parametrize("kind,normalize", CASES)
def test_materialise_matches_oracle(ctx, oracle, kind, normalize):
    import panovlm_amd as pv
    rng = np.random.default_rng(100 + kind * 2 + int(normalize))
    aa, t, ref, nei, rows, off = _build(rng, kind)
    w = 1.7
    rs = pv.ResidualSet.upload(ctx, kind, rows, off, ref, nei, flags=1 if normalize else 0, weight=w)
    ctx.set_poses(aa, t)
    r, J = rs.eval(jac=True)
    rid, nid = synth.expand_ids(off, ref, nei)
    orows = synth.oracle_rows(kind, rows, w)
    ro, Jo = oracle.evaluate(kind, orows, rid, nid, aa, t, normalize=normalize, extended=True)     # exact value of upstream's formula
    rd, Jd = oracle.evaluate(kind, orows, rid, nid, aa, t, normalize=normalize)                    # what upstream computes in double
    assert r.shape == ro.shape and J.shape == Jo.shape
    # decisions: the early-out of the *_Angle functors must fall on the same blocks as upstream's; blocks whose branch
    # distance sits on the threshold (within 1e-12 relative) are counted — none may hide in a tolerance
    if kind in (1, 3):
        dis = oracle.branch_distance(kind, orows, rid, nid, aa, t)
        on_threshold = np.abs(dis - 1e-3) <= 1e-15
        assert on_threshold.sum() == 0
        assert np.array_equal(r == 0, dis < 1e-3)
    assert np.array_equal(rd == 0, r == 0) and np.array_equal(ro == 0, r == 0)
    assert np.all(np.abs(r - ro) <= RTOL * np.abs(ro) + ATOL), (np.abs(r - ro) / (RTOL * np.abs(ro) + ATOL)).max()
    scale = np.maximum(np.abs(Jo).max(axis=1, keepdims=True), 1e-9)
    assert np.all(np.abs(J - Jo) <= RTOL * scale), (np.abs(J - Jo) / (RTOL * scale)).max()
    # upstream's own double evaluation against the same arbiter: it is the looser of the two near r -> 0 (acos of a
    # cosine next to 1 loses eps / r, its derivative eps / r^2) and only passes with that conditioning term added —
    # the allowance round 1 had put on the GPU comparison belongs here
    cond = 8 * 2.2e-16 / np.maximum(np.abs(ro), 1e-300)
    assert np.all(np.abs(rd - ro) <= RTOL * np.abs(ro) + ATOL + cond)
    with np.errstate(over="ignore"):
        assert np.all(np.abs(Jd - Jo) <= (RTOL + (cond / np.maximum(np.abs(ro), 1e-300))[:, None]) * scale)
    # the GPU is at least as close to the exact value as upstream's double arithmetic, up to rounding noise
    assert np.abs(J - Jo).max() <= max(10 * np.abs(Jd - Jo).max(), 1e-10 * scale.max())
    # cost-only evaluation returns identical residuals
    r2, J2 = rs.eval(jac=False)
    assert J2 is None and np.array_equal(r, r2)
    rs.close()


@pytest.mark.parametrize("kind,normalize", CASES)
@pytest.mark.parametrize("loss", [0, 1])
def test_fused_pair_blocks_match_oracle(ctx, oracle, kind, normalize, loss):
    import panovlm_amd as pv
    rng = np.random.default_rng(200 + kind * 2 + int(normalize))
    aa, t, ref, nei, rows, off = _build(rng, kind)
    a = 2 * np.pi / 180 if kind in (1, 3, 4, 5) else 0.2
    rs = pv.ResidualSet.upload(ctx, kind, rows, off, ref, nei, flags=1 if normalize else 0, weight=0.9)
    ctx.set_poses(aa, t)
    blocks = rs.pair_blocks(loss, a)
    rid, nid = synth.expand_ids(off, ref, nei)
    ro, Jo = oracle.evaluate(kind, synth.oracle_rows(kind, rows, 0.9), rid, nid, aa, t, normalize=normalize)
    expect = synth.pair_blocks_from_jacobian(ro, Jo, off, loss, a)
    scale = np.maximum(np.abs(expect).max(axis=1, keepdims=True), 1e-12)
    assert np.all(np.abs(blocks - expect) <= RTOL * scale), (np.abs(blocks - expect) / scale).max()
    # determinism: bit-identical on a second run
    assert np.array_equal(blocks, rs.pair_blocks(loss, a))
    rs.close()


@pytest.mark.parametrize("kind", [1, 3])
def test_early_out_threshold_is_decided_like_upstream(ctx, oracle, kind):
    """Blocks constructed to straddle `dis < 1e-3` (CostFunction.h:680-684, :893-897): the point is moved along the plane
    normal / away from the line so that the branch distance is 1e-3 (1 +- d) for d from 1e-3 down to 1e-12.  At every
    margin >= 1e-9 the GPU takes upstream's branch on every block; below that the blocks are counted as on-threshold."""
    import panovlm_amd as pv
    rng = np.random.default_rng(900 + kind)
    F = 5
    aa, t = synth.random_poses(rng, F)
    ref = np.array([0, 1, 2, 3], np.int32); nei = np.array([1, 2, 3, 4], np.int32)
    rows, off = synth.random_resset(rng, kind, aa, t, ref, nei, np.full(4, 60))
    rid, nid = synth.expand_ids(off, ref, nei)
    flips = 0
    for margin in (1e-3, 1e-6, 1e-9, 1e-12):
        for side in (-1.0, 1.0):
            target = 1e-3 * (1.0 + side * margin)
            rr = rows.copy()
            # move P_n so that the branch distance becomes `target`: for both functors the distance is linear along the
            # offset direction, so two Newton corrections in extended precision land within 1e-16
            for _ in range(4):
                d = oracle.branch_distance(kind, synth.oracle_rows(kind, rr), rid, nid, aa, t)
                eps = 1e-7
                grad = np.zeros((len(rr), 3))
                for k in range(3):
                    pert = rr.copy(); pert[:, k] += eps
                    grad[:, k] = (oracle.branch_distance(kind, synth.oracle_rows(kind, pert), rid, nid, aa, t) - d) / eps
                g2 = np.maximum((grad * grad).sum(1), 1e-30)
                rr[:, :3] += ((target - d) / g2)[:, None] * grad
            d = oracle.branch_distance(kind, synth.oracle_rows(kind, rr), rid, nid, aa, t)
            ok = np.abs(d - target) <= 1e-3 * margin * 0.25            # rows that really sit on the wanted side of the threshold
            rs = pv.ResidualSet.upload(ctx, kind, rr, off, ref, nei, flags=1)
            ctx.set_poses(aa, t)
            r, _ = rs.eval(jac=False)
            rd, _ = oracle.evaluate(kind, synth.oracle_rows(kind, rr), rid, nid, aa, t, normalize=True, jac=False)
            rs.close()
            if margin >= 1e-9:
                assert ok.sum() >= 200
                assert np.array_equal((r == 0)[ok], (d < 1e-3)[ok]) and np.array_equal((r == 0)[ok], (rd == 0)[ok])
            else:
                flips += int(((r == 0) != (rd == 0)).sum())
    assert flips <= 240        # reported, not hidden: at a 1e-15 m margin GPU and CPU may round the distance to different sides


def test_normal_equations_packed(ctx, oracle):
    import panovlm_amd as pv
    rng = np.random.default_rng(7)
    F, P = 9, 40
    aa, t = synth.random_poses(rng, F)
    ref, nei = synth.random_pairs(rng, F, P)
    counts = rng.integers(1, 200, size=P)
    rows, off = synth.random_resset(rng, 1, aa, t, ref, nei, counts)
    rs = pv.ResidualSet.upload(ctx, 1, rows, off, ref, nei, flags=1)
    ctx.set_poses(aa, t)
    up = sorted({(min(a, b), max(a, b)) for a, b in zip(ref.tolist(), nei.tolist())})
    neq = pv.NormalEq(ctx, F, [u[0] for u in up], [u[1] for u in up])
    a = 2 * np.pi / 180
    packed = neq.accumulate(rs, 1, a)
    Hd, Ho, g, cost = neq.unpack(packed)
    rid, nid = synth.expand_ids(off, ref, nei)
    ro, Jo = oracle.evaluate(1, synth.oracle_rows(1, rows), rid, nid, aa, t, normalize=True)
    w, half_rho = synth.huber_weights(ro, 1, a)
    H = np.zeros((F * 6, F * 6)); gg = np.zeros(F * 6)
    for i in range(len(ro)):
        idx = np.concatenate([np.arange(6) + 6 * rid[i], np.arange(6) + 6 * nid[i]])
        H[np.ix_(idx, idx)] += w[i] * np.outer(Jo[i], Jo[i])
        gg[idx] += w[i] * Jo[i] * ro[i]
    sc = np.abs(H).max()
    for i in range(F):
        assert np.allclose(Hd[i], H[6 * i:6 * i + 6, 6 * i:6 * i + 6], rtol=0, atol=1e-9 * sc)
        assert np.allclose(g[i], gg[6 * i:6 * i + 6], rtol=0, atol=1e-9 * np.abs(gg).max())
    for u, (i, j) in enumerate(up):
        assert np.allclose(Ho[u], H[6 * i:6 * i + 6, 6 * j:6 * j + 6], rtol=0, atol=1e-9 * sc)
    assert np.isclose(cost, half_rho.sum(), rtol=1e-10)
    # accumulate a second time on top (+=)
    packed2 = neq.accumulate(rs, 1, a, packed=packed.copy())
    assert np.allclose(packed2, 2 * packed, rtol=1e-12)
    # the queued form: two structures, two sets, ONE synchronisation — each landing buffer equals the synchronous call bit for bit
    # (what an LM driver does with the point-to-plane and the line-to-line blocks of one problem); both evaluation forms of the
    # fused kernel (workgroup- and wave-per-chunk, chosen by segment length) are behind it
    rows_b, off_b = synth.random_resset(rng, 1, aa, t, ref, nei, rng.integers(3000, 9000, size=P))
    rs_b = pv.ResidualSet.upload(ctx, 1, rows_b, off_b, ref, nei, flags=1)
    neq_b = pv.NormalEq(ctx, F, [u[0] for u in up], [u[1] for u in up])
    want_a, want_b = neq.accumulate(rs, 1, a), neq_b.accumulate(rs_b, 1, a)
    got_a = np.full(neq.size, np.nan); got_b = np.full(neq_b.size, np.nan)
    neq.accumulate_async(rs, got_a, 1, a)
    neq_b.accumulate_async(rs_b, got_b, 1, a)
    ctx.synchronize()
    assert np.array_equal(got_a, want_a) and np.array_equal(got_b, want_b) and not np.array_equal(got_a, got_b)
    # ... and the summed form: a set uploaded in ANOTHER pose numbering (ids shifted by 3) is renumbered into the common one
    # (pvlm_resset_set_pose_ids) and both sets are added on the device into one buffer = the sum of the two synchronous results
    rs_c = pv.ResidualSet.upload(ctx, 1, rows_b, off_b, ref + 3, nei + 3, flags=1)
    rs_c.set_pose_ids(ref, nei)
    po, pr, pn, _ = rs_c.download()
    assert np.array_equal(pr, ref) and np.array_equal(pn, nei)
    neq_c = pv.NormalEq(ctx, F, [u[0] for u in up], [u[1] for u in up])
    summed = np.full(neq.size, np.nan)
    pv.NormalEq.accumulate_sets(ctx, [neq, neq_c], [rs, rs_c], [1, 1], [a, a], summed)
    ctx.synchronize()
    assert np.array_equal(summed, want_a + want_b)          # one addition per entry, the same one the host would make
    with pytest.raises(pv.PvlmError):                       # different structures cannot share a buffer
        other = pv.NormalEq(ctx, F + 1, [u[0] for u in up], [u[1] for u in up])
        try:
            pv.NormalEq.accumulate_sets(ctx, [neq, other], [rs, rs_c], [1, 1], [a, a], summed)
        finally:
            other.close()
    neq_c.close(); rs_c.close(); neq_b.close(); rs_b.close()


def test_errors_are_loud(ctx):
    import panovlm_amd as pv
    rows = np.zeros((2, 7)); rows[:, 3] = 1
    rs = pv.ResidualSet.upload(ctx, 0, rows, [0, 2], [0], [5])
    ctx.set_poses(np.zeros((2, 3)), np.zeros((2, 3)))
    with pytest.raises(pv.PvlmError):
        rs.eval()           # pose id 5 outside the table
    with pytest.raises(pv.PvlmError):
        pv.ResidualSet.upload(ctx, 0, rows, [0, 1], [0], [1])   # offsets do not span n
