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

Run from the repo root:  python tests/golden/make_golden.py
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


if __name__ == "__main__":
    known_answers()
    assoc_small()
