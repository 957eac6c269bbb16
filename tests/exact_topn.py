"""Oracle-INDEPENDENT expectations for the association scan's top-N and the kinship counts (test infrastructure).

Nothing here calls the oracle or the product: integers, fractions and NumPy's IEEE doubles only.

Why it can be exact. With integer phenotypes y_i and S * max|y| < 2^24, every float32 addition the reference performs in
calculate_kmer_score (src/kmers_multiple_databases.cpp:327-363) is exact in ANY order, and so is its sequential float32
column sum (:288-295). Then
    yigi = sum_{i present} y_i                      (an integer)
    r    = N * yigi - N1 * sum(y)                   (an integer, exact in double)
    score = r * r / (N * N1 - N1 * N1)              (:359-361)
and where r^2 < 2^53 the product r * r is exact too, so the reference's score IS the rational r^2 / d rounded ONCE to a
double - float(Fraction(r * r, d)), correctly rounded by Python. The expected top-N of a column is then: rows that pass the
MAC predicate (:119), sorted by that double; BestAssociationsHeap (src/best_associations_heap.cpp:43-59) keeps the N largest
whatever libstdc++'s heap layout does, PROVIDED the N-th and (N+1)-th doubles differ (asserted); entries with equal doubles
INSIDE the top-N all stay, only their relative pop order depends on the heap's history - compare() below therefore orders
equal scores by row on both sides. The number of effective pushes (add_association calls that change the heap) depends on
the multiset of scores only (strict '>' against the current minimum), so it is simulated here with heapq.
"""
import hashlib
import heapq
from fractions import Fraction

import numpy as np

from helpers import splitmix64, synth_rows_numpy


# ---- the cases: everything is regenerated from these integers ------------------------------------------------------
#  name: S, P, n_rows, topn, table seed, phenotype seed, ymax[, bumps: columns re-drawn because their first draw had a tie
#  across the top-N boundary (found by tests/golden/make_golden.py --find-bumps)]
CASES = {
    # 241 accessions (BASELINE configs[0] shape), 24 columns: small enough for the push-count simulation
    "s241_p24": dict(S=241, P=24, n_rows=200_000, topn=1000, seed=11, yseed=3, ymax=900),
    # BASELINE configs[1] shape: 1024 x 101, the pipeline's top-10001
    "s1024_p101": dict(S=1024, P=101, n_rows=200_000, topn=10001, seed=12, yseed=5, ymax=600, bumps={21: 1}),
    # BASELINE configs[2] shape: 1135 accessions (2 whole 512-sample groups + 1 quarter group in the block-scaled filter)
    "s1135_p40": dict(S=1135, P=40, n_rows=200_000, topn=2001, seed=13, yseed=7, ymax=600),
    # BASELINE configs[3] sample count: 2048 x 64 (several LDS groups per launch)
    "s2048_p64": dict(S=2048, P=64, n_rows=200_000, topn=1001, seed=14, yseed=9, ymax=400),
    # 1, 2 and 4 columns: the narrow filter (FP4 x FP8 block-scaled MFMA), the pipeline's top-10001
    "s1024_p1": dict(S=1024, P=1, n_rows=200_000, topn=10001, seed=15, yseed=11, ymax=600),
    "s1135_p2": dict(S=1135, P=2, n_rows=200_000, topn=10001, seed=16, yseed=13, ymax=600),
    "s2048_p4": dict(S=2048, P=4, n_rows=200_000, topn=4001, seed=17, yseed=15, ymax=400),
}


class BoundaryTie(Exception):
    """A column whose N-th and (N+1)-th scores are the same double: which one stays depends on the heap's layout."""

    def __init__(self, column):
        super().__init__("tie across the top-N boundary in column %d: bump its seed (CASES[...]['bumps'])" % column)
        self.column = column


def int_phenotypes(S, P, ymax, yseed, bumps=None):
    """P x S integers in [-ymax, ymax], a pure function of (yseed, column, accession, bump of the column).
    bumps: {column: k} - re-draws of single columns (make_golden.py finds the few columns that need one: a tie across the
    top-N boundary cannot be decided without the heap's layout)."""
    bumps = bumps or {}
    col = np.arange(P, dtype=np.uint64)
    bump = np.asarray([bumps.get(j, 0) for j in range(P)], np.uint64)
    idx = (col * np.uint64(S))[:, None] + np.arange(S, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        h = splitmix64(idx + np.uint64(yseed) * np.uint64(0x9E3779B97F4A7C15) + (bump * np.uint64(0xD6E8FEB86659FD93))[:, None])
    return (h % np.uint64(2 * ymax + 1)).astype(np.int64) - ymax


def make_inputs(name):
    c = CASES[name]
    rows = synth_rows_numpy(0, c["n_rows"], c["S"], c["seed"])  # kmer = row + 1, ~6 % of the rows fail the MAC filter
    Yi = int_phenotypes(c["S"], c["P"], c["ymax"], c["yseed"], c.get("bumps"))
    assert c["S"] * c["ymax"] < (1 << 24)  # every float32 partial sum is an exactly representable integer
    mac = max(int(np.ceil(c["S"] * 0.05)), 5)  # associate_kmers.cpp:99-103 with maf 0.05, mac 5
    return rows, Yi, mac, c["topn"]


def _bits(rows, S):
    w = np.ascontiguousarray(rows[:, 1:])
    b = np.unpackbits(w.view(np.uint8), axis=1, bitorder="little")
    return b[:, :S]


def exact_scores(rows, S, Yi, mac, block=16384):
    """r (int64, n x P), d (int64, n), kept (bool, n): the integers behind every score."""
    n = len(rows)
    P = Yi.shape[0]
    Yt = np.ascontiguousarray(Yi.T.astype(np.float32))  # integers below 2^24: exact
    sumy = Yi.sum(axis=1).astype(np.int64)
    r = np.empty((n, P), np.int64)
    N1 = np.empty(n, np.int64)
    for a in range(0, n, block):
        b = _bits(rows[a:a + block], S)
        n1 = b.sum(axis=1, dtype=np.int64)
        yg = b.astype(np.float32) @ Yt  # float32 GEMM of integers whose partial sums stay below 2^24: exact in any order
        yg64 = yg.astype(np.int64)
        assert (yg64.astype(np.float32) == yg).all()
        N1[a:a + block] = n1
        r[a:a + block] = S * yg64 - n1[:, None] * sumy[None, :]
    d = S * N1 - N1 * N1
    kept = (N1 >= mac) & (N1 <= S - mac)
    return r, d, kept


def expected_topn(rows, S, Yi, mac, topn, margin=64):
    """Per column: (file rows, kmers, scores) in pop order (ascending score, equal scores by ascending row), and `tested`.
    Raises if a case cannot be decided without knowing the heap's layout (a tie across the top-N boundary)."""
    r, d, kept = exact_scores(rows, S, Yi, mac)
    idx_kept = np.nonzero(kept)[0]
    rk, dk = r[idx_kept], d[idx_kept]
    out = []
    for j in range(Yi.shape[0]):
        rj = rk[:, j]
        approx = rj.astype(np.float64) ** 2 / dk.astype(np.float64)  # within a few ulp of the true value
        m = min(len(rj), topn + margin)
        cand = np.argpartition(-approx, m - 1)[:m] if m < len(rj) else np.arange(len(rj))
        cand = cand[np.argsort(-approx[cand], kind="stable")]
        if m < len(rj) and topn <= len(rj):
            # the candidates certainly contain the true top-N: the first one left out is clearly below the N-th
            assert approx[cand[-1]] < approx[cand[min(topn, m) - 1]] * (1 - 1e-9) or m == len(rj), "raise the margin"
        ex = []
        for t in cand:
            rr = int(rj[t])
            assert rr * rr < (1 << 53), "r^2 must be exact in a double (lower ymax)"
            ex.append((Fraction(rr * rr, int(dk[t])), int(idx_kept[t])))
        ex.sort(key=lambda e: (-e[0], e[1]))
        take = ex[:topn]
        sc = [float(e[0]) for e in ex]  # correctly rounded, once
        if len(ex) > topn:
            if not sc[topn - 1] > sc[topn]:
                raise BoundaryTie(j)
        take.sort(key=lambda e: (float(e[0]), e[1]))  # pop order: ascending score, ties by row
        rw = np.asarray([e[1] for e in take], np.uint64)
        out.append((rw, rows[rw.astype(np.int64), 0].copy(), np.asarray([float(e[0]) for e in take], np.float64)))
    return out, int(kept.sum())


def effective_pushes(rows, S, Yi, mac, topn):
    """How many add_association calls change a heap, summed over the columns (a function of the score multiset only)."""
    r, d, kept = exact_scores(rows, S, Yi, mac)
    idx = np.nonzero(kept)[0]
    sc = (r[idx].astype(np.float64) * r[idx].astype(np.float64)) / d[idx].astype(np.float64)[:, None]  # IEEE r*r, then /
    total = 0
    for j in range(Yi.shape[0]):
        h = []
        col = sc[:, j].tolist()
        for v in col:
            if len(h) < topn:
                heapq.heappush(h, v)
                total += 1
            elif v > h[0]:
                heapq.heapreplace(h, v)
                total += 1
    return total


def canonical(rows, kmers, scores):
    """Pop order made layout-independent: equal scores ordered by row."""
    o = np.lexsort((rows, scores))
    return rows[o], kmers[o], scores[o]


def compare(got, exp):
    """got / exp: (rows, kmers, scores) of one column."""
    gr, gk, gs = got
    er, ek, es = exp
    assert len(gr) == len(er), (len(gr), len(er))
    assert (np.diff(gs) >= 0).all(), "pop order is not ascending"
    gr, gk, gs = canonical(np.asarray(gr, np.uint64), np.asarray(gk, np.uint64), np.asarray(gs, np.float64))
    assert gs.tobytes() == es.tobytes(), "scores differ from the exact rationals"
    assert (gr == er).all(), "row identities differ"
    assert (gk == ek).all(), "k-mers differ"


def digest(exp_cols):
    """One sha256 per column over (rows, scores) - what tests/golden/exact_topn.json pins."""
    return [hashlib.sha256(e[0].astype("<u8").tobytes() + e[2].astype("<f8").tobytes()).hexdigest() for e in exp_cols]


# ---- kinship: the closed form K_ij = n - c_i - c_j + 2 c_ij over the rows passing the MAF predicate ---------------
def kinship_closed_form(rows, S_f, maf=0.05):
    """update_emma_kinshhip_calculation (src/kmers_multiple_databases.cpp:418-438) counts, over rows with
    ceil(S_f * maf) <= popcount <= S_f - that, K[i][j] += 1 ^ g_i ^ g_j = [g_i == g_j]; with c_ij = sum g_i g_j that is
    n - c_i - c_j + 2 c_ij (integers; the diagonal is n). Returns (K int64 S_f x S_f, n)."""
    mc = int(np.ceil(S_f * maf))
    b = _bits(rows, S_f)
    n1 = b.sum(axis=1, dtype=np.int64)
    use = (n1 >= mc) & (n1 <= S_f - mc)
    g = b[use].astype(np.float64)  # 0/1, sums below 2^53: exact
    n = int(use.sum())
    c = (g.T @ g).astype(np.int64)
    ci = np.diag(c)
    K = n - ci[:, None] - ci[None, :] + 2 * c
    return K, n, mc
