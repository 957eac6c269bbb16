"""CPU tests of the oracle itself (no GPU): the three statements of the reference's float32
order agree bit-for-bit, the heap equals an independent restatement of libstdc++'s algorithms,
hand-derived known answers hold, and the committed golden vectors are reproduced.

The reference ships no tests or golden vectors for this path and cannot be built in this image
(see oracle/oracle.cpp header), so the goldens here are ORACLE-generated regression vectors, plus
exact-arithmetic known answers that do not depend on any implementation.
"""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

from oracle import binding as ob
from oracle import oracle_np as onp
from helpers import random_table, phenotypes

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("S_f,S,reorder", [(5, 5, False), (64, 64, False), (127, 127, False), (128, 128, False),
                                           (129, 129, False), (241, 241, False), (241, 200, True), (1027, 1027, False),
                                           (300, 77, True)])
def test_three_statements_agree(S_f, S, reorder):
    rows = random_table(300, S_f, seed=S_f * 7 + S)
    rng = np.random.default_rng(S)
    col = rng.permutation(S_f)[:S].astype(np.uint64) if reorder else np.arange(S, dtype=np.uint64)
    Y = phenotypes(S, 2, seed=S + 1)
    mac = onp.min_count(S, 0.05, 2)
    sc0, kept0 = ob.scores_dense(rows, S_f, col, Y, mac, which=0)
    sc1, kept1 = ob.scores_dense(rows, S_f, col, Y, mac, which=1)
    assert (kept0 == kept1).all()
    assert sc0.tobytes() == sc1.tobytes()  # scalar statement == SSE statement, bit for bit
    g, n1, keep = onp.mac_filter(rows, col, mac)
    assert (keep == kept0).all()
    for j in range(Y.shape[0]):
        sn = onp.scores(g, n1, Y[j], mac)
        sn = np.where(keep, sn, 0.0)
        assert sn.tobytes() == sc0[j].tobytes()  # NumPy statement == C++ statements


def test_known_answers_exact_arithmetic():
    """Small-integer phenotypes make every float32 operation exact, so the score is the rational
    (N*sum(y_i g_i) - N1*sum(y))^2 / (N*N1 - N1^2), evaluated here with fractions."""
    cases = json.load(open(os.path.join(GOLD, "known_answers.json")))
    assert len(cases) >= 8
    for c in cases:
        S = c["S"]
        y = np.asarray(c["y"], np.float32)
        bits = c["bits"]
        W = (S + 63) // 64
        words = [0] * W
        for i, b in enumerate(bits):
            if b:
                words[i // 64] |= 1 << (i % 64)
        rows = np.asarray([[1] + words], dtype=np.uint64)
        col = np.arange(S, dtype=np.uint64)
        sc, kept = ob.scores_dense(rows, S, col, y[None, :], c["mac"], which=0)
        # independent expectation
        N, N1 = S, sum(bits)
        if N1 >= c["mac"] and N1 <= S - c["mac"]:
            yg = sum(Fraction(int(v)) for v, b in zip(c["y"], bits) if b)
            r = N * yg - N1 * sum(Fraction(int(v)) for v in c["y"])
            exp = float(r * r / (N * N1 - N1 * N1))
            assert kept[0]
        else:
            exp = 0.0
            assert not kept[0]
        assert sc[0, 0] == exp == c["expected"]


def test_permute_scores_is_the_sse_lane_order():
    y = np.arange(1, 257, dtype=np.float32)
    R = np.zeros(256, np.float32)
    s = ob.lib().orc_prepare_scores(y, 256, 4, R)
    for b in range(2):
        for sx in range(32):
            for l in range(4):
                assert R[128 * b + 4 * sx + l] == y[128 * b + 32 * l + 31 - sx]
    assert s == np.float32(256 * 257 / 2)
    assert onp.permuted_sum(y) == s


def _tie_stream(n, seed, levels):
    rng = np.random.default_rng(seed)
    scores = rng.integers(0, levels, size=n).astype(np.float64) / 4.0
    return np.arange(n, dtype=np.uint64) + 100, scores, np.arange(n, dtype=np.uint64)


@pytest.mark.parametrize("N,n,levels", [(1, 50, 3), (7, 400, 4), (64, 3000, 10), (100, 90, 5), (33, 2000, 2)])
def test_heap_matches_libstdcxx_restatement(N, n, levels):
    """std::priority_queue (oracle.cpp) vs the pure-Python push_heap/pop_heap restatement, on
    streams with massive ties: identical survivors, pop order and ranks."""
    k, s, r = _tie_stream(n, seed=N * 1000 + n, levels=levels)
    h = ob.Heap(N)
    h.add_many(k, s, r)
    p = onp.BestHeap(N)
    for i in range(n):
        p.add(int(k[i]), float(s[i]), int(r[i]))
    hk, hs, hr = h.pop_all()
    pp = p.pop_all()
    assert [int(x) for x in hk] == [e[0] for e in pp]
    assert [float(x) for x in hs] == [e[1] for e in pp]
    assert [int(x) for x in hr] == [e[2] for e in pp]
    ok, ork, orow = h.output_list()
    pl = p.output_list()
    assert [(int(a), int(b), int(c)) for a, b, c in zip(ok, ork, orow)] == pl
    assert h.insertions == n
    assert (np.diff(hs) >= 0).all()


def test_heap_nan_and_strict_greater():
    h = ob.Heap(2)
    h.add_many([1, 2, 3, 4, 5], [1.0, 2.0, 1.0, float("nan"), 2.0], [0, 1, 2, 3, 4])
    k, s, r = h.pop_all()
    # 3 (score 1.0 == lowest) is rejected: strict '>'; NaN never displaces; 5 (2.0 > 1.0) replaces 1
    assert sorted(int(x) for x in k) == [2, 5]


def test_associate_matches_bruteforce_without_ties():
    S = 130
    rows = random_table(3000, S, seed=11)
    col = np.arange(S, dtype=np.uint64)
    Y = phenotypes(S, 3, seed=5)
    mac = onp.min_count(S, 0.05, 5)
    dense, kept = ob.scores_dense(rows, S, col, Y, mac)
    res = ob.associate(rows, S, col, Y, 50, mac, batch_size=700, threads=2)
    assert res["tested"] == int(kept.sum())
    for j in range(Y.shape[0]):
        sc = dense[j][kept]
        idx = np.nonzero(kept)[0]
        order = np.argsort(sc, kind="stable")[-50:]
        assert len(np.unique(sc[order])) == 50  # no ties in this draw
        got = res["per_pheno"][j]
        assert (got["file_row"] == idx[order]).all()
        assert got["score"].tobytes() == sc[order].tobytes()
        assert (got["kmer"] == rows[idx[order], 0]).all()


def test_associate_batch_size_and_threads_do_not_matter():
    S = 70
    rows = random_table(2500, S, seed=3, dup_frac=0.5)
    col = np.arange(S, dtype=np.uint64)
    Y = phenotypes(S, 2, seed=8, binary=True)
    mac = 4
    a = ob.associate(rows, S, col, Y, 40, mac, batch_size=10_000_000, threads=1)
    b = ob.associate(rows, S, col, Y, 40, mac, batch_size=97, threads=3)
    for j in range(3):
        for key in ("kmer", "score", "file_row"):
            assert a["per_pheno"][j][key].tobytes() == b["per_pheno"][j][key].tobytes()
    assert a["tested"] == b["tested"]


def test_kinship_loop_equals_closed_form():
    S = 77
    rows = random_table(800, S, seed=21)
    mc = int(np.ceil(S * 0.05))
    K, n = ob.kinship(rows, S, mc)
    K2, n2 = onp.kinship(rows, S, 0.05)
    assert n == n2 and (K == K2).all()
    txt = ob.kinship_text(K, n).decode()
    lines = txt.strip("\n").split("\n")
    assert len(lines) == S and all(len(l.split("\t")) == S for l in lines)
    assert lines[3].split("\t")[3] == "1"


def test_table_roundtrip_and_guards(tmp_path):
    S = 70
    rows = random_table(20, S, seed=2)
    names = ["acc%d" % i for i in range(S)]
    base = str(tmp_path / "t")
    onp.write_table(base, names, 31, rows[:, 0], rows[:, 1:])
    n2, r2 = onp.read_table(base, 31)
    assert n2 == names and (r2 == rows).all()
    with pytest.raises(ValueError, match="Kmer length"):
        onp.read_table(base, 25)
    raw = bytearray(open(base + ".table", "rb").read())
    open(base + ".table", "wb").write(raw[:-8])
    with pytest.raises(ValueError, match="size of file"):
        onp.read_table(base, 31)
    raw[0] = 0
    open(base + ".table", "wb").write(raw)
    with pytest.raises(ValueError, match="Incorrect prefix"):
        onp.read_table(base, 31)
    with pytest.raises(ValueError, match="Couldn't find"):
        onp.column_map(names, ["nope"])
    with pytest.raises(ValueError, match="same name"):
        onp.column_map(names + ["acc1"], ["acc1"])


def test_bits2kmer():
    assert ob.bits2kmer(0b00011011, 4) == "ACGT"
    assert onp.bits2kmer(0b00011011, 4) == "ACGT"
    assert ob.bits2kmer(3, 31) == "A" * 30 + "T"


def test_golden_regression_vectors():
    """tests/golden/assoc_small.npz was produced by tests/golden/make_golden.py from this oracle;
    it pins the oracle (and, on the GPU box, the HIP path) against silent drift."""
    g = np.load(os.path.join(GOLD, "assoc_small.npz"))
    rows, col, Y = g["rows"], g["col"], g["Y"]
    S_f, mac, topn = int(g["S_f"]), int(g["mac"]), int(g["topn"])
    dense, kept = ob.scores_dense(rows, S_f, col, Y, mac)
    assert dense.tobytes() == g["dense"].tobytes()
    assert (kept == g["kept"]).all()
    res = ob.associate(rows, S_f, col, Y, topn, mac)
    for j in range(Y.shape[0]):
        assert (res["per_pheno"][j]["kmer"] == g["top_kmer"][j]).all()
        assert res["per_pheno"][j]["score"].tobytes() == g["top_score"][j].tobytes()
        assert (res["per_pheno"][j]["file_row"] == g["top_row"][j]).all()
    K, n = ob.kinship(rows, S_f, int(g["kin_min_count"]))
    assert n == int(g["kin_n"]) and (K == g["kin_K"]).all()


# ---- the oracle against implementation-independent arithmetic at production size --------------------------------------
def _exact_fixture():
    return json.load(open(os.path.join(GOLD, "exact_topn.json")))


@pytest.mark.parametrize("name", ["s241_p24", "s1024_p101", "s1135_p40", "s2048_p64", "s1024_p1", "s1135_p2", "s2048_p4"])
def test_oracle_topn_equals_exact_rationals(name):
    """tests/exact_topn.py: 200 k synthetic rows, integer phenotypes - every float32 add of the reference is exact, the score
    is a rational rounded once, the top-N is decided by integer arithmetic (no oracle, no product code). The committed
    fixture (tests/golden/exact_topn.json) pins that computation; the oracle's heaps must hold exactly those entries, with
    those score bytes, in ascending pop order (entries of EQUAL score may pop in either order: that depends on the heap's
    history and is pinned elsewhere, against libstdc++ itself)."""
    import exact_topn as ex
    c = ex.CASES[name]
    fx = _exact_fixture()["cases"][name]
    rows, Yi, mac, topn = ex.make_inputs(name)
    exp, tested = ex.expected_topn(rows, c["S"], Yi, mac, topn)
    assert ex.digest(exp) == fx["sha256"] and tested == fx["tested"] and mac == fx["mac"]  # the fixture pins the expectation
    for j in (0, c["P"] - 1):
        f = fx["column_%d" % j]
        assert [int(v) for v in exp[j][0][:64]] == f["rows_lowest"] and [float(v).hex() for v in exp[j][2][-64:]] == f["score_hex_best"]
    res = ob.associate(rows, c["S"], np.arange(c["S"], dtype=np.uint64), Yi.astype(np.float32), topn, mac, threads=4)
    assert res["tested"] == tested
    for j in range(c["P"]):
        o = res["per_pheno"][j]
        ex.compare((o["file_row"], o["kmer"], o["score"]), exp[j])
    if "effective_pushes" in fx:
        assert ex.effective_pushes(rows, c["S"], Yi, mac, topn) == fx["effective_pushes"] == res["pushes"]


@pytest.mark.parametrize("name", ["kin_s241", "kin_s1135"])
def test_oracle_kinship_equals_closed_form_fixture(name):
    """The kinship loop (1 ^ g_i ^ g_j per pair and row) against K_ij = n - c_i - c_j + 2 c_ij from NumPy integers."""
    import hashlib
    import exact_topn as ex
    fx = _exact_fixture()["kinship"][name]
    n_rows = fx["n_rows"] if name == "kin_s241" else 6000  # (the loop is S_f^2 / 2 per row: a slice of the large case)
    rows = ex.synth_rows_numpy(0, fx["n_rows"], fx["S_f"], fx["seed"])
    K, n, mc = ex.kinship_closed_form(rows, fx["S_f"])
    assert n == fx["n_used"] and mc == fx["min_count"]
    assert hashlib.sha256(K.astype("<u8").tobytes()).hexdigest() == fx["sha256"]
    K2, n2, _ = ex.kinship_closed_form(rows[:n_rows], fx["S_f"])
    Ko, no = ob.kinship(rows[:n_rows], fx["S_f"], mc)
    assert no == n2
    iu = np.tril_indices(fx["S_f"], -1)  # the loop fills j < i; what it leaves elsewhere is the formatter's business
    assert (np.asarray(Ko, np.int64)[iu] == K2[iu]).all()
