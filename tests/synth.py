"""Seeded synthetic inputs shared by the tests, smoke() and bench.py (numpy only)."""
import numpy as np


def rodrigues(aa):
    aa = np.asarray(aa, np.float64)
    th = np.linalg.norm(aa)
    K = np.array([[0, -aa[2], aa[1]], [aa[2], 0, -aa[0]], [-aa[1], aa[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def random_poses(rng, F, rot=0.3, trans=2.0):
    return rng.normal(size=(F, 3)) * rot, rng.normal(size=(F, 3)) * trans


def random_pairs(rng, F, P):
    ref = rng.integers(0, F, size=P)
    nei = (ref + rng.integers(1, F, size=P)) % F
    return ref.astype(np.int32), nei.astype(np.int32)


def p_ref(aa, t, r, n, P):
    return rodrigues(aa[r]) @ rodrigues(-aa[n]) @ (P - t[n]) + t[r]


def random_resset(rng, kind, aa, t, ref, nei, counts, offsets_scale=(1e-4, 5e-3, 0.05, 0.5)):
    """ICP-like residual rows for each (ref[p], nei[p]) segment; returns rows (n x stride), pair_offsets."""
    rows = []
    off = [0]
    for p, c in enumerate(counts):
        for _ in range(int(c)):
            P = rng.normal(size=3) * 4.0
            Pr = p_ref(aa, t, ref[p], nei[p], P)
            o = rng.choice(offsets_scale) * rng.choice([-1.0, 1.0])
            if kind in (0, 1):
                n = rng.normal(size=3); n /= np.linalg.norm(n)
                rows.append(np.concatenate([P, n, [-(n @ Pr) + o]]))
            elif kind in (2, 3):
                d = rng.normal(size=3); d /= np.linalg.norm(d)
                u = rng.normal(size=3); u -= (u @ d) * d; u /= np.linalg.norm(u)
                A0 = Pr + u * abs(o) + d * rng.normal()
                rows.append(np.concatenate([P, A0 + 0.1 * d, A0 - 0.1 * d]))
            elif kind == 4:
                a = rng.normal(size=3) * 3; b = a + rng.normal(size=3)
                ac = p_ref(aa, t, ref[p], nei[p], a); bc = p_ref(aa, t, ref[p], nei[p], b)
                nrm = np.cross(ac, bc); nrm /= np.linalg.norm(nrm)
                pert = nrm + rng.normal(size=3) * abs(o)
                rows.append(np.concatenate([pert * rng.uniform(0.5, 2.0), a, b, [rng.uniform(0.5, 2.0)]]))
            else:
                m = rng.normal(size=3) * 3
                mc = p_ref(aa, t, ref[p], nei[p], m)
                n = np.cross(mc, rng.normal(size=3)); n /= np.linalg.norm(n)
                n = n + rng.normal(size=3) * 0.05
                mid = mc / np.linalg.norm(mc) + rng.normal(size=3) * rng.choice([0.01, 0.3])
                rows.append(np.concatenate([n * rng.uniform(0.5, 2), [0.0], m, mid, [rng.uniform(0.005, 0.4)], [rng.uniform(0.5, 2.0)]]))
        off.append(len(rows))
    stride = {0: 7, 1: 7, 2: 9, 3: 9, 4: 10, 5: 12}[kind]
    rows = np.array(rows, np.float64).reshape(-1, stride)
    return rows, np.array(off, np.int64)


def oracle_rows(kind, rows, weight=1.0):
    """Converts ABI rows to the oracle's row layout (per-row weight column for kinds 0..3)."""
    if kind in (0, 1, 2, 3):
        return np.concatenate([rows, np.full((rows.shape[0], 1), weight)], axis=1)
    return rows


def expand_ids(pair_offsets, ref, nei):
    counts = np.diff(pair_offsets)
    return np.repeat(ref, counts).astype(np.int32), np.repeat(nei, counts).astype(np.int32)


def huber_weights(r, loss, a):
    s = r * r
    if loss == 0:
        return np.ones_like(r), 0.5 * s
    rho1 = np.ones_like(r); rho = s.copy()
    m = s > a * a
    rr = np.sqrt(s[m])
    rho[m] = 2 * a * rr - a * a
    rho1[m] = a / rr
    return rho1, 0.5 * rho


def pair_blocks_from_jacobian(r, J, pair_offsets, loss, a):
    """Reference assembly of the 121-double pair blocks from materialised r, J (numpy)."""
    P = len(pair_offsets) - 1
    out = np.zeros((P, 121))
    w, half_rho = huber_weights(r, loss, a)
    for p in range(P):
        s, e = pair_offsets[p], pair_offsets[p + 1]
        Jp = J[s:e]; rp = r[s:e]; wp = w[s:e]
        H = (Jp * wp[:, None]).T @ Jp
        g = (Jp * wp[:, None]).T @ rp
        out[p, 0:36] = H[0:6, 0:6].reshape(-1)
        out[p, 36:72] = H[0:6, 6:12].reshape(-1)
        out[p, 72:108] = H[6:12, 6:12].reshape(-1)
        out[p, 108:120] = g
        out[p, 120] = half_rho[s:e].sum()
    return out
