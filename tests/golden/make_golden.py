"""Generates the committed fixtures in tests/golden/.

  known_answers.json : implementation-independent known answers. Phenotypes are small integers, so
                       every float32 add the reference performs is exact and the score equals the
                       rational (N*sum(y_i g_i) - N1*sum(y))^2 / (N*N1 - N1^2); `expected` is that
                       rational rounded once to float64 (Python's Fraction -> float is correctly
                       rounded), computed WITHOUT calling any oracle or product code.
  assoc_small.npz    : ORACLE-generated regression vectors (the reference has no tests or fixtures
                       for this path and cannot be built here — see oracle/oracle.cpp). A 600-row,
                       150-accession table with 40 % duplicated presence/absence patterns, a subset of
                       131 phenotyped accessions in shuffled order, one binary phenotype + 3 permutations.

  exact_topn.json    : implementation-independent top-N answers at production size (tests/exact_topn.py: integer
                       phenotypes, exact rationals rounded once, 200 k-row tables regenerated from seeds): per case
                       and column a sha256 over (file rows, score bytes) in pop order, the 64 weakest and 64 best entries of each
                       case's first and last column, the MAC-passing row count, and for the smallest case the number of
                       effective heap pushes. Computed WITHOUT calling any oracle or product code. Also the kinship
                       counts of two synthetic tables from the closed form K_ij = n - c_i - c_j + 2 c_ij (sha256).

Run from the repo root:  python tests/golden/make_golden.py            (all fixtures)
                         python tests/golden/make_golden.py --find-bumps  (columns of tests/exact_topn.py CASES whose
                         draw has a tie across the top-N boundary, to be listed in the case's "bumps")
"""
import json
import os
import sys
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def known_answers():
    rng = np.random.default_rng(20240601)
    cases = []
    for S, mac in [(4, 1), (5, 1), (64, 3), (127, 6), (128, 6), (129, 6), (200, 10), (241, 13), (300, 15), (130, 70)]:
        for rep in range(3):
            y = [int(v) for v in rng.integers(-40, 41, size=S)]
            f = rng.uniform(0.05, 0.95)
            bits = [int(v) for v in (rng.random(S) < f)]
            N, N1 = S, sum(bits)
            if N1 >= mac and N1 <= S - mac:
                yg = sum(Fraction(v) for v, b in zip(y, bits) if b)
                r = N * yg - N1 * sum(Fraction(v) for v in y)
                exp = float(r * r / (N * N1 - N1 * N1))
            else:
                exp = 0.0
            cases.append(dict(S=S, mac=mac, y=y, bits=bits, expected=exp))
    json.dump(cases, open(os.path.join(HERE, "known_answers.json"), "w"))
    print("known_answers.json:", len(cases), "cases")


def assoc_small():
    from helpers import random_table, phenotypes
    from oracle import binding as ob
    from oracle import oracle_np as onp
    S_f, S = 150, 131
    rows = random_table(600, S_f, seed=99, dup_frac=0.4)
    col = np.random.default_rng(5).permutation(S_f)[:S].astype(np.uint64)
    Y = phenotypes(S, 3, seed=17, binary=True)
    mac = onp.min_count(S, 0.05, 5)
    topn = 25
    dense, kept = ob.scores_dense(rows, S_f, col, Y, mac)
    res = ob.associate(rows, S_f, col, Y, topn, mac)
    kin_mc = int(np.ceil(S_f * 0.05))
    K, n = ob.kinship(rows, S_f, kin_mc)
    np.savez_compressed(
        os.path.join(HERE, "assoc_small.npz"), rows=rows, col=col, Y=Y, S_f=S_f, mac=mac, topn=topn, dense=dense,
        kept=kept, top_kmer=np.stack([r["kmer"] for r in res["per_pheno"]]),
        top_score=np.stack([r["score"] for r in res["per_pheno"]]),
        top_row=np.stack([r["file_row"] for r in res["per_pheno"]]), tested=res["tested"], kin_min_count=kin_mc,
        kin_K=K, kin_n=n)
    print("assoc_small.npz: tested", res["tested"], "kin n", n)


KINSHIP_CASES = {"kin_s241": dict(S_f=241, n_rows=60_000, seed=31), "kin_s1135": dict(S_f=1135, n_rows=40_000, seed=32)}


def exact_topn_fixture():
    import hashlib
    import exact_topn as ex
    out = {"cases": {}, "kinship": {}}
    for name, c in ex.CASES.items():
        rows, Yi, mac, topn = ex.make_inputs(name)
        exp, tested = ex.expected_topn(rows, c["S"], Yi, mac, topn)
        e = dict(tested=tested, mac=mac, sha256=ex.digest(exp), ties_inside=int(sum((np.diff(x[2]) == 0).sum() for x in exp)))
        for j in sorted({0, c["P"] - 1}):
            lo, hi = slice(0, 64), slice(-64, None)  # the weakest and the best 64 entries, readable; the sha256 pins all of them
            e["column_%d" % j] = dict(rows_lowest=[int(v) for v in exp[j][0][lo]], score_hex_lowest=[float(v).hex() for v in exp[j][2][lo]],
                                      rows_best=[int(v) for v in exp[j][0][hi]], score_hex_best=[float(v).hex() for v in exp[j][2][hi]])
        if name == "s241_p24":
            e["effective_pushes"] = ex.effective_pushes(rows, c["S"], Yi, mac, topn)
        out["cases"][name] = e
        print("exact_topn:", name, "tested", tested, "ties inside the top-N", e["ties_inside"])
    for name, c in KINSHIP_CASES.items():
        rows = ex.synth_rows_numpy(0, c["n_rows"], c["S_f"], c["seed"])
        K, n, mc = ex.kinship_closed_form(rows, c["S_f"])
        out["kinship"][name] = dict(c, n_used=n, min_count=mc, sha256=hashlib.sha256(K.astype("<u8").tobytes()).hexdigest(),
                                    first_row=[int(v) for v in K[0, :8]])
        print("exact kinship:", name, "rows used", n)
    json.dump(out, open(os.path.join(HERE, "exact_topn.json"), "w"))


def find_bumps():
    import exact_topn as ex
    for name, c in ex.CASES.items():
        rows = ex.synth_rows_numpy(0, c["n_rows"], c["S"], c["seed"])
        mac = max(int(np.ceil(c["S"] * 0.05)), 5)
        bumps = dict(c.get("bumps", {}))
        for j in range(c["P"]):
            while True:
                Yi = ex.int_phenotypes(c["S"], c["P"], c["ymax"], c["yseed"], bumps)[j:j + 1]
                try:
                    ex.expected_topn(rows, c["S"], Yi, mac, c["topn"])
                    break
                except ex.BoundaryTie:
                    bumps[j] = bumps.get(j, 0) + 1
        print(name, "bumps =", bumps, flush=True)


if __name__ == "__main__":
    if "--find-bumps" in sys.argv:
        find_bumps()
        sys.exit(0)
    known_answers()
    assoc_small()
    exact_topn_fixture()
