"""CPU: self-consistency of the VLQ oracle restatement (there is no CPU reference
for this path: PARITY UNPINNED, see oracle/vlq_oracle.cpp)."""
import numpy as np
import pytest

from oracle import pyoracle


def make_vlq(seed=0, d=32, nlist=24, M=8, nbits=6, nedge=6, nlambda=16, nb=3000):
    rng = np.random.default_rng(seed)
    centres = rng.random((40, d)).astype(np.float32)
    def gen(n):
        return (centres[rng.integers(0, 40, n)] + 0.08 * rng.standard_normal((n, d))).astype(np.float32)
    xt = gen(2000)
    coarse = xt[rng.permutation(2000)[:nlist]].copy()
    v = pyoracle.OracleVLQ(d, nlist, M, nbits, nedge, nlambda, coarse)
    line, lam = v.assign(xt)
    # 1-D codebook over the training lambdas: quantiles (training is outside the path)
    v.lambda_info = np.quantile(lam, (np.arange(nlambda) + 0.5) / nlambda).astype(np.float32)
    lb = v.quantize_lambda(lam)
    res = v.residuals(xt, line, lb)
    ksub, dsub = 1 << nbits, d // M
    pq = np.stack([res[rng.permutation(2000)[:ksub], m * dsub:(m + 1) * dsub] for m in range(M)])
    v.pq_centroids = np.ascontiguousarray(pq, np.float32)
    xb = gen(nb)
    v.add(xb)
    return v, xb, gen(40)


def decode(v, pos):
    line = np.searchsorted(v.line_off, pos, side="right") - 1
    c, e = line // v.nedge, line % v.nedge
    s = v.edge_info[c, e]
    l = v.lambda_info[v.lambdas[pos]].astype(np.float64)
    anchor = (1 - l) * v.coarse[c].astype(np.float64) + l * v.coarse[s].astype(np.float64)
    r = np.concatenate([v.pq_centroids[m, v.codes[pos, m]] for m in range(v.M)]).astype(np.float64)
    return anchor + r


def test_graph_excludes_self_and_is_sorted():
    v, _, _ = make_vlq()
    for i in range(v.nlist):
        assert i not in v.edge_info[i]
        assert np.all(np.diff(v.edge_dist[i]) >= 0)
        d2 = ((v.coarse[i] - v.coarse[v.edge_info[i]]) ** 2).sum(1)
        assert np.allclose(d2, v.edge_dist[i], rtol=1e-4, atol=1e-4)


def test_assignment_minimises_distance_to_line():
    v, xb, _ = make_vlq()
    line, lam = v.assign(xb[:200])
    near = v.nearest(xb[:200])
    for i in range(200):
        A = near[i]
        assert line[i] // v.nedge == A
        x = xb[i].astype(np.float64)
        best = None
        for e in range(v.nedge):
            c, s = v.coarse[A].astype(np.float64), v.coarse[v.edge_info[A, e]].astype(np.float64)
            l = np.dot(x - c, s - c) / np.dot(s - c, s - c)
            d2 = ((x - (c + l * (s - c))) ** 2).sum()
            inside = 0 <= l <= 1
            if best is None or (inside, -d2) > (best[0], -best[1]):
                best = (inside, d2, e, l)
        # same choice up to float rounding of near-equal candidates
        e_or = line[i] % v.nedge
        c, s = v.coarse[A].astype(np.float64), v.coarse[v.edge_info[A, e_or]].astype(np.float64)
        l_or = np.dot(x - c, s - c) / np.dot(s - c, s - c)
        d_or = ((x - (c + l_or * (s - c))) ** 2).sum()
        assert d_or <= best[1] * (1 + 1e-4) + 1e-6 or (0 <= l_or <= 1) > best[0]
        assert abs(lam[i] - l_or) < 1e-3


def test_search_distances_match_decoded_vectors():
    """dist = |q - anchor - r|^2 - |q|^2 recomputed in float64 from the stored codes."""
    v, xb, xq = make_vlq()
    D, I = v.search(xq, nprobe=8, w1=24, k=10)
    pos_of = {int(i): p for p, i in enumerate(v.ids)}
    for qi in range(xq.shape[0]):
        q = xq[qi].astype(np.float64)
        for j in range(10):
            if I[qi, j] < 0:
                continue
            y = decode(v, pos_of[int(I[qi, j])])
            ref = ((q - y) ** 2).sum() - (q ** 2).sum()
            assert abs(D[qi, j] - ref) <= 1e-3 * max(1.0, abs(ref)), (qi, j, D[qi, j], ref)
        assert np.all(np.diff(D[qi][I[qi] >= 0]) >= 0)


def test_search_with_all_lines_is_exhaustive_over_the_scanned_codes():
    """With every line selected the result is the true top-k of the formula over all codes
    (lines are capped at 1024 codes, none is that long here)."""
    v, xb, xq = make_vlq(nb=800)
    D, I, lines = v.search(xq[:5], nprobe=v.nlist, w1=v.nlist * v.nedge, k=5, return_lines=True)
    assert v.last_ncode == 5 * 800
    for qi in range(5):
        q = xq[qi].astype(np.float64)
        allv = np.array([((q - decode(v, p)) ** 2).sum() - (q ** 2).sum() for p in range(800)])
        best = np.sort(allv)[:5]
        assert np.allclose(D[qi], best, rtol=1e-3, atol=1e-3)
