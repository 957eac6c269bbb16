"""CPU tests of the product's host side (no GPU, no compute calls): the C-ABI library loads and
exports every symbol include/kgwas.h declares, the .table/.names/.pheno readers and their error
behaviour, the BestAssociationsHeap mirror, the cross-shard merge, the PLINK and kinship writers,
the synthetic-row generator's host twin, the CLIs' argument handling — each against the oracle."""
import os
import re
import subprocess
import sys

import ctypes as C
import numpy as np
import pytest

import kmersgwas_amd as kg
from kmersgwas_amd import capi
from oracle import binding as ob
from oracle import oracle_np as onp
from helpers import random_table, phenotypes, synth_rows_numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "kmersgwas_amd", "bin")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "kgwas.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(kgwas_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(capi.SYMBOLS), "capi.SYMBOLS must list exactly what kgwas.h declares"
    for s in declared:
        assert hasattr(capi.lib, s), "libkgwas.so does not export " + s
    assert capi.lib.kgwas_version() >= 100
    # one HIP runtime per process: the library itself must not pull a second one in
    out = subprocess.check_output(["readelf", "-d", capi.LIB_PATH]).decode()
    assert "libamdhip64" not in out


def _write_table(tmp_path, n_rows=50, S=70, k=31, seed=2):
    rows = random_table(n_rows, S, seed=seed)
    names = ["acc%03d" % i for i in range(S)]
    base = str(tmp_path / "t")
    onp.write_table(base, names, k, rows[:, 0], rows[:, 1:])
    return base, names, rows


def test_table_reader_and_guards(tmp_path):
    base, names, rows = _write_table(tmp_path)
    t = kg.KmersTable(base, 31)
    assert (t.n_acc, t.n_rows, t.words_per_row, t.kmer_len) == (70, 50, 2, 31)
    assert t.names == names
    assert (t.read_rows(0, 50) == rows).all()
    assert (t.read_rows(7, 5) == rows[7:12]).all()
    with pytest.raises(kg.KgwasError) as e:
        t.read_rows(48, 3)
    assert e.value.code == capi.KGWAS_ERR_ARG
    assert (t.column_map(["acc005", "acc000", "acc069"]) == [5, 0, 69]).all()
    with pytest.raises(kg.KgwasError, match="Couldn't find path for DB: nope") as e:
        t.column_map(["acc001", "nope"])
    assert e.value.code == capi.KGWAS_ERR_FORMAT
    t.close()
    assert kg.KmersTable(base, 0).kmer_len == 31  # k = 0 skips the check
    # the reference's logic_error messages (src/kmers_multiple_databases.cpp:65-93)
    with pytest.raises(kg.KgwasError, match="Kmer length not as defined in class"):
        kg.KmersTable(base, 25)
    raw = bytearray(open(base + ".table", "rb").read())
    open(base + ".table", "wb").write(raw[:-8])
    with pytest.raises(kg.KgwasError, match="size of file not valid"):
        kg.KmersTable(base, 31)
    open(base + ".table", "wb").write(raw[:16])
    with pytest.raises(kg.KgwasError, match="Kmer table size is too small"):
        kg.KmersTable(base, 31)
    bad = bytearray(raw)
    bad[0] = 0
    open(base + ".table", "wb").write(bad)
    with pytest.raises(kg.KgwasError, match="Incorrect prefix"):
        kg.KmersTable(base, 31)
    open(base + ".table", "wb").write(raw)
    open(base + ".names", "a").write("extra\n")
    with pytest.raises(kg.KgwasError, match="Number of accession in file not as defined in class"):
        kg.KmersTable(base, 31)
    with pytest.raises(kg.KgwasError, match="Couldn't open kmer table file") as e:
        kg.KmersTable(str(tmp_path / "missing"), 31)
    assert e.value.code == capi.KGWAS_ERR_IO


def test_duplicate_accession_in_names(tmp_path):
    base, names, rows = _write_table(tmp_path)
    names2 = list(names)
    names2[3] = names2[9]
    open(base + ".names", "w").write("\n".join(names2) + "\n")
    t = kg.KmersTable(base, 31)
    with pytest.raises(kg.KgwasError, match="Two DBs with the same name! acc009"):
        t.column_map(["acc009"])


@pytest.mark.parametrize("fname", ["resistence.pheno", "FT10.pheno"])
def test_phenotype_loader_on_the_reference_examples(fname):
    path = os.path.join(GOLD, fname)
    p = kg.Phenotypes(path)
    names, acc, Y = onp.load_phenotypes(path)
    assert p.names == names == ["phenotype_value"]
    assert p.accessions == acc
    assert p.Y.tobytes() == Y.tobytes()
    assert len(acc) == (241 if fname.startswith("res") else 1162)


def test_phenotype_loader_multi_column_and_errors(tmp_path):
    f = tmp_path / "p.tsv"
    f.write_text("accession_id\tphenotype_value\tP1\tP2\nA\t1.5\t-2e-3\t7\nB\t0.1\t3.25\t1e10\nC\t71.6666666667\t0\t-0")
    p = kg.Phenotypes(str(f))
    names, acc, Y = onp.load_phenotypes(str(f))
    assert p.names == names == ["phenotype_value", "P1", "P2"] and p.accessions == acc == ["A", "B", "C"]
    assert p.Y.tobytes() == Y.tobytes()
    f.write_text("accession_id\tv\nA\t1\nB\n")
    with pytest.raises(kg.KgwasError, match="same number of fields") as e:
        kg.Phenotypes(str(f))
    assert e.value.code == capi.KGWAS_ERR_FORMAT
    f.write_text("accession_id\tv\nA\tabc\n")
    with pytest.raises(kg.KgwasError):
        kg.Phenotypes(str(f))
    with pytest.raises(kg.KgwasError) as e:
        kg.Phenotypes(str(tmp_path / "none.tsv"))
    assert e.value.code == capi.KGWAS_ERR_IO


def test_min_count_matches_the_reference_rule():
    for S, maf, mac in [(241, 0.05, 5), (1024, 0.05, 5), (2048, 0.05, 5), (1135, 0.05, 5), (20, 0.05, 5), (100, 0.0, 0),
                        (1000, 0.051, 3)]:
        assert kg.min_count(S, maf, mac) == onp.min_count(S, maf, mac) == int(ob.lib().orc_min_count(S, maf, mac))
    assert kg.min_count(241, 0.05, 5) == 13 and kg.min_count(1024, 0.05, 5) == 52 and kg.min_count(2048, 0.05, 5) == 103


@pytest.mark.parametrize("N,n,levels", [(1, 60, 3), (2, 100, 3), (3, 200, 2), (10, 500, 4), (64, 4000, 9), (200, 150, 5), (33, 3000, 2),
                                        (1000, 30000, 40), (1001, 30000, 2000),
                                        (70001, 400000, 3000)])  # (> 65 536 entries: the hole walks prefetch the descendants three levels down)
@pytest.mark.parametrize("flavour", ["nan", "plain", "negative", "distinct", "one_tie", "low_tie", "mid_ties"])
def test_heap_mirror_equals_oracle_heap_under_ties(N, n, levels, flavour):
    """(plain: scores in +0 .. +inf, the heap compares their bit patterns as integers; nan / negative: it must notice
    and compare as doubles; distinct: no two scores equal - the pop sequence is then produced by sorting, not popping;
    one_tie: distinct but for ONE pair of equal scores among the largest - sorting must give up and pop for real;
    low_tie / mid_ties: distinct but for a pair of equal scores in the lowest tenth of the entries that stay / three pairs
    spread over their lower half - the heap is popped for real up to its last tied score and sorted from there)"""
    rng = np.random.default_rng(N * 31 + n)
    k = np.arange(n, dtype=np.uint64) + 7
    s = rng.integers(0, levels, size=n).astype(np.float64) / 8.0
    if flavour in ("distinct", "one_tie", "low_tie", "mid_ties"):
        s = rng.permutation(n).astype(np.float64) * 0.37 + rng.random(n) * 0.1  # all different
        top = np.argsort(s)[-min(N, n):]  # the entries that stay, ascending
        if flavour == "one_tie" and n >= 4:
            s[top[0]] = s[top[-1]]  # the weakest entry that stays gets the best one's score
        if flavour == "low_tie" and len(top) >= 4:
            a = len(top) // 10
            s[top[a]] = s[top[a + 1]]
        if flavour == "mid_ties" and len(top) >= 16:
            for f in (0.05, 0.3, 0.5):
                a = int(f * len(top))
                s[top[a]] = s[top[a + 1]]
    if flavour == "nan":
        s[rng.random(n) < 0.01] = np.nan
    elif flavour == "negative":
        s -= 0.25
        s[rng.random(n) < 0.05] = -0.0
    elif flavour not in ("low_tie", "mid_ties"):  # (several +inf are ties at the very top: everything would be popped for real)
        s[rng.random(n) < 0.01] = np.inf
    r = np.arange(n, dtype=np.uint64) * 3
    h = kg.BestAssociationsHeap(N)
    h.add_associations(k[: n // 2], s[: n // 2], r[: n // 2])
    h.add_associations(k[n // 2:], s[n // 2:], r[n // 2:])
    o = ob.Heap(N)
    o.add_many(k, s, r)
    for a, b in zip(h.pop_all(), o.pop_all()):
        assert a.tobytes() == b.tobytes()
    for a, b in zip(h.get_kmers_for_output(), o.output_list()):
        assert (a == b).all()
    assert len(h) == min(N, n)
    # get_rows_sorted_indices (:135-147), output_to_file / _with_scores (:65-90): the rows ascending; the pops as raw bytes
    ok, os_, orow = o.pop_all()
    assert (h.get_rows_sorted_indices() == np.sort(orow)).all()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        h.output_to_file(os.path.join(d, "k"))
        h.output_to_file(os.path.join(d, "ks"), with_scores=True)
        assert open(os.path.join(d, "k"), "rb").read() == ok.tobytes()
        both = np.empty((len(ok), 2), np.uint64)
        both[:, 0] = ok
        both[:, 1] = os_.view(np.uint64)
        assert open(os.path.join(d, "ks"), "rb").read() == both.tobytes()


@pytest.mark.parametrize("by_ref,detach", [(1, -1), (0, -1), (1, 2), (1, 0)])
@pytest.mark.parametrize("N,n_chunks,per,flavour", [
    (500, 12, 900, "distinct"), (500, 12, 900, "tie_inside"), (500, 12, 900, "tie_at_boundary"), (500, 12, 900, "tie_below"),
    (500, 12, 900, "nan"), (500, 12, 900, "negative"), (500, 12, 900, "placeholders"), (500, 12, 900, "levels"),
    (5000, 3, 700, "distinct"), (1, 5, 50, "distinct"), (64, 40, 300, "late_tie"),
    (300, 30, 800, "no_thresholds"), (300, 30, 800, "no_thresholds_tie")])
def test_select_mode_decision_equals_the_oracle_heap(N, n_chunks, per, flavour, by_ref, detach):
    """scan_lazy.cpp on the host alone (kgwas_select_check): a column's records arrive in chunks, each with a threshold that is
    some lower bound of the N-th largest score so far; the column is finished by SELECTION iff its N + 1 largest scores are
    pairwise distinct and no score is NaN or negative, by a replay of its log otherwise - and its lists equal the oracle's
    std::priority_queue either way. Logs by reference and by copy, detached from the caller's arrays mid-stream, pools pruned
    with the thresholds, -inf placeholders (the narrow filter's survivors that are no candidates), heaps that never fill."""
    rng = np.random.default_rng(N * 7 + n_chunks * 13 + per + len(flavour))
    n = n_chunks * per
    score = rng.permutation(n).astype(np.float64) * 0.37 + rng.random(n) * 0.1  # all different, all positive
    order = np.argsort(score)
    top = order[-min(N, n):]  # ascending: top[0] is the N-th largest
    expect_selected = 1
    if flavour == "tie_inside":
        score[top[len(top) // 2]] = score[top[-1]]
        expect_selected = 0
    elif flavour == "tie_at_boundary" and n > N:
        score[order[-N - 1]] = score[top[0]]  # the (N + 1)-th largest equals the N-th
        expect_selected = 0
    elif flavour == "tie_below" and n > N + 3:
        score[order[-N - 3]] = score[order[-N - 2]]  # two equal scores that are not among the N + 1 largest: no matter
    elif flavour == "late_tie":
        last = np.arange(n - per, n)
        a = last[np.argmax(score[last])]
        score[a] = score[top[-1]] if top[-1] != a else score[top[-2]]  # a row of the LAST chunk repeats one of the best scores
        expect_selected = 0
    elif flavour == "nan":
        score[rng.integers(0, n)] = np.nan
        expect_selected = 0
    elif flavour == "negative":
        score[rng.integers(0, n)] = -0.0
        expect_selected = 0
    elif flavour == "levels":
        score = rng.integers(0, 40, size=n).astype(np.float64) / 8.0
        expect_selected = 0
    elif flavour == "no_thresholds_tie":
        score[top[3]] = score[top[7]]
        expect_selected = 0
    if flavour == "placeholders":
        score[rng.random(n) < 0.2] = -np.inf
    kmer = (np.arange(n, dtype=np.uint64) * 5 + 11)
    row_in_chunk = np.tile(np.arange(per, dtype=np.uint32) * 2, n_chunks)
    chunk_row0 = (np.arange(n_chunks, dtype=np.uint64) * (2 * per + 3))
    chunk_n = np.full(n_chunks, per, np.uint64)
    # thresholds: the N-th largest real score up to the end of each chunk, lowered by a random bit (0 while there are fewer than N)
    thr = np.zeros(n_chunks)
    for c in range(n_chunks):
        seen = score[: (c + 1) * per]
        seen = seen[np.isfinite(seen) & (seen >= 0)]
        # (no_thresholds: the device never raised a bound - the pool is cut by selection alone, LazyCol::compact)
        if len(seen) >= N and flavour not in ("nan", "negative", "no_thresholds", "no_thresholds_tie"):
            thr[c] = np.partition(seen, len(seen) - N)[len(seen) - N] * (1.0 - 0.004 * rng.random())
    real = score != -np.inf
    rows = (np.repeat(chunk_row0, per) + row_in_chunk)
    o = ob.Heap(N)
    o.add_many(kmer[real], score[real], rows[real])
    ek, es, er = o.pop_all()
    sel = C.c_int(-1)
    out_n = C.c_uint64(0)
    ok_ = np.zeros(N, np.uint64); os_ = np.zeros(N, np.float64); or_ = np.zeros(N, np.uint64)
    from kmersgwas_amd import capi
    rc = capi.lib.kgwas_select_check(N, n_chunks, chunk_n.ctypes.data, chunk_row0.ctypes.data, thr.ctypes.data, score.ctypes.data, kmer.ctypes.data,
                                     row_in_chunk.ctypes.data, by_ref, detach, C.byref(sel), ok_.ctypes.data, os_.ctypes.data, or_.ctypes.data, C.byref(out_n))
    assert rc == 0, capi.last_error() if hasattr(capi, "last_error") else rc
    m = out_n.value
    assert m == len(ek)
    assert ok_[:m].tobytes() == ek.tobytes() and os_[:m].tobytes() == es.tobytes() and or_[:m].tobytes() == er.tobytes()
    assert sel.value == expect_selected, (flavour, sel.value)


@pytest.mark.parametrize("by_ref,detach", [(1, 0), (1, 2), (0, -1), (1, -1)])
@pytest.mark.parametrize("empty", [(0,), (0, 1), (2,), (0, 3, 5)])
def test_select_mode_with_empty_chunks(by_ref, detach, empty):
    """Chunks that ship no record for a column (a zero-length first chunk with the logs detached behind it was undefined
    behaviour in LazyCol::detach - advisor, round 5: `segs.back()` of an empty vector): lists equal the oracle heap's."""
    rng = np.random.default_rng(17 + len(empty))
    N, n_chunks = 40, 6
    sizes = np.array([0 if c in empty else 90 for c in range(n_chunks)], np.uint64)
    n = int(sizes.sum())
    score = rng.permutation(n).astype(np.float64) + 0.5
    kmer = np.arange(n, dtype=np.uint64) + 100
    row_in_chunk = np.concatenate([np.arange(int(k), dtype=np.uint32) for k in sizes]) if n else np.zeros(0, np.uint32)
    chunk_row0 = np.arange(n_chunks, dtype=np.uint64) * 1000
    thr = np.zeros(n_chunks)
    rows = np.concatenate([chunk_row0[c] + np.arange(int(sizes[c]), dtype=np.uint64) for c in range(n_chunks)])
    o = ob.Heap(N)
    o.add_many(kmer, score, rows)
    ek, es, er = o.pop_all()
    sel, out_n = C.c_int(-1), C.c_uint64(0)
    ok_ = np.zeros(N, np.uint64); os_ = np.zeros(N, np.float64); or_ = np.zeros(N, np.uint64)
    rc = capi.lib.kgwas_select_check(N, n_chunks, sizes.ctypes.data, chunk_row0.ctypes.data, thr.ctypes.data, score.ctypes.data, kmer.ctypes.data,
                                     row_in_chunk.ctypes.data, by_ref, detach, C.byref(sel), ok_.ctypes.data, os_.ctypes.data, or_.ctypes.data, C.byref(out_n))
    assert rc == 0
    m = out_n.value
    assert m == len(ek) == N and sel.value == 1
    assert ok_[:m].tobytes() == ek.tobytes() and os_[:m].tobytes() == es.tobytes() and or_[:m].tobytes() == er.tobytes()


def _python_history(kmer, score, row, N):
    """Effective pushes of a shard-local heap (what kgwas_scan_history returns), via the pure-Python heap."""
    h = onp.BestHeap(N)
    hist = ([], [], [])
    for i in range(len(kmer)):
        before = (len(h.q), h.lowest)
        h.add(int(kmer[i]), float(score[i]), int(row[i]))
        changed = len(h.q) != before[0] or (before[0] == N and score[i] > before[1])
        if changed:
            hist[0].append(kmer[i]); hist[1].append(score[i]); hist[2].append(row[i])
    return (np.asarray(hist[0], np.uint64), np.asarray(hist[1], np.float64), np.asarray(hist[2], np.uint64))


def test_merge_shards_equals_single_pass():
    """The union of shard-local effective pushes, replayed shard by shard, reproduces the single heap."""
    rng = np.random.default_rng(3)
    P, N, M = 4, 25, 3000
    kmer = np.arange(M, dtype=np.uint64) + 1000
    row = np.arange(M, dtype=np.uint64)
    scores = [rng.integers(0, 40, size=M).astype(np.float64) / 4 for _ in range(P)]  # many ties
    cuts = [0, 700, 701, 2200, M]
    shards = []
    for g in range(len(cuts) - 1):
        lo, hi = cuts[g], cuts[g + 1]
        shards.append([_python_history(kmer[lo:hi], scores[j][lo:hi], row[lo:hi], N) for j in range(P)])
    heaps = kg.merge_shards(N, shards, threads=2)
    for j in range(P):
        o = ob.Heap(N)
        o.add_many(kmer, scores[j], row)
        for a, b in zip(heaps[j].pop_all(), o.pop_all()):
            assert a.tobytes() == b.tobytes()


def test_plink_writer_matches_oracle_bytes(tmp_path):
    S_f, S = 150, 131
    rows = random_table(400, S_f, seed=12)
    names = ["s%d" % i for i in range(S_f)]
    base = str(tmp_path / "tab")
    onp.write_table(base, names, 31, rows[:, 0], rows[:, 1:])
    col = np.random.default_rng(4).permutation(S_f)[:S].astype(np.uint64)
    acc = [names[c] for c in col]
    y = np.asarray([71.6666666667, 1e-05, 100000, 1234567, -0.0, 0.5, 3] + list(np.linspace(-2, 2, S - 7)), np.float32)
    rng = np.random.default_rng(6)
    pick = rng.choice(400, size=60, replace=False)
    kmer_pop, row_pop = rows[pick, 0], pick.astype(np.uint64)
    t = kg.KmersTable(base, 31)
    kg.write_plink(str(tmp_path / "prod"), t, col, acc, y, kmer_pop, row_pop)
    ob.write_plink(str(tmp_path / "orc"), rows, S_f, col, acc, y, 31, kmer_pop, row_pop)
    for ext in (".bed", ".bim", ".fam"):
        a = open(str(tmp_path / "prod") + ext, "rb").read()
        b = open(str(tmp_path / "orc") + ext, "rb").read()
        assert a == b, ext
    bed = open(str(tmp_path / "prod.bed"), "rb").read()
    assert bed[:3] == b"\x6c\x1b\x01" and len(bed) == 3 + 60 * ((S + 3) // 4)
    bim = open(str(tmp_path / "prod.bim")).read().split("\n")
    assert bim[0].startswith("0\t") and re.match(r"^0\t[ACGT]{31}_\d+\t0\t0\t0\t1$", bim[0])


@pytest.mark.parametrize("S_f,S,reorder,threads", [(150, 131, True, 3), (150, 150, False, 0), (150, 131, False, 2), (64, 64, False, 1),
                                                   (70, 5, True, 4), (1135, 1135, False, 0)])
def test_plink_writer_for_all_columns_matches_oracle_bytes(tmp_path, S_f, S, reorder, threads):
    """kgwas_write_plink_many (the batched, multi-threaded pass 2 of the command-line tool) against the oracle's per-column
    writer: shuffled subsets and leading columns (the word-wise expansion; fewer phenotyped accessions than the table has,
    so set bits beyond them must not leak into the last byte), accession counts that are not multiples of 4, columns that
    share winners, winners next to each other and far apart (coalesced and separate reads), an empty column, more union
    rows than one block's pieces."""
    n_rows = 3000
    rows = random_table(n_rows, S_f, seed=S_f + S)
    names = ["s%d" % i for i in range(S_f)]
    base = str(tmp_path / "tab")
    onp.write_table(base, names, 31, rows[:, 0], rows[:, 1:])
    rng = np.random.default_rng(S)
    col = rng.permutation(S_f)[:S].astype(np.uint64) if reorder else np.arange(S, dtype=np.uint64)
    acc = [names[c] for c in col]
    P = 7
    Y = rng.standard_normal((P, S)).astype(np.float32)
    Y[0, :3] = [71.6666666667, 1e-05, -0.0]
    picks = [rng.choice(n_rows, size=n, replace=False) for n in (900, 1, 0, 2000, 300, 900, 57)]
    picks[5] = picks[0][::-1].copy()                    # the same winners in another pop order (other ranks)
    picks[4] = np.arange(1000, 1300)[rng.permutation(300)]  # a run of neighbouring rows
    t = kg.KmersTable(base, 31)
    outs = [str(tmp_path / ("prod%d" % j)) for j in range(P)]
    kg.write_plink_many(outs, t, col, acc, Y, [rows[p, 0] for p in picks], [p.astype(np.uint64) for p in picks], threads=threads)
    for j in range(P):
        orc = str(tmp_path / ("orc%d" % j))
        ob.write_plink(orc, rows, S_f, col, acc, Y[j], 31, rows[picks[j], 0], picks[j].astype(np.uint64))
        for ext in (".bed", ".bim", ".fam"):
            assert open(outs[j] + ext, "rb").read() == open(orc + ext, "rb").read(), (j, ext)
    with pytest.raises(kg.KgwasError):
        kg.write_plink_many(outs[:1], t, col, acc, Y[:1], [rows[:1, 0]], [np.asarray([n_rows], np.uint64)])  # row beyond the table
    # a failure on the library's helper threads (files that cannot be created, several threads at work) comes back as the
    # call's error - not as a terminated process - and the library goes on working
    bad = [str(tmp_path / "no_such_dir" / ("x%d" % j)) for j in range(P)]
    with pytest.raises(kg.KgwasError):
        kg.write_plink_many(bad, t, col, acc, Y, [rows[p, 0] for p in picks], [p.astype(np.uint64) for p in picks], threads=4)
    again = [str(tmp_path / ("again%d" % j)) for j in range(P)]
    kg.write_plink_many(again, t, col, acc, Y, [rows[p, 0] for p in picks], [p.astype(np.uint64) for p in picks], threads=4)
    assert open(again[3] + ".bed", "rb").read() == open(outs[3] + ".bed", "rb").read()
    t.close()


def test_kinship_text_and_from_partials():
    S = 50
    rows = random_table(300, S, seed=5)
    K, n = ob.kinship(rows, S, 3)
    g = onp.unpack_bits(rows, np.arange(S, dtype=np.uint64)).astype(np.int64)
    n1 = g.sum(axis=1)
    g = g[(n1 >= 3) & (n1 <= S - 3)]
    H = ((g[:, :, None] ^ g[:, None, :]).sum(axis=0)).astype(np.uint64)  # Hamming distances
    K2 = kg.kinship_from_partials(H, len(g))
    assert len(g) == n and (K2 == K).all()
    assert kg.kinship_format(K2, n) == ob.kinship_text(K, n)


def test_kinship_text_covers_the_formats_of_ostream_double():
    """`os << double` (src/emma_kinship_kmers.cpp:104-109) is %g with six digits: fixed and exponent forms, trailing zeros
    dropped, rounding at the sixth digit - the library formats with std::to_chars on several threads; the oracle with an
    ostream. Ratios k / n from every decade down to 1e-12, around the 1e-5 switch to exponents, and round ones."""
    S = 160  # several threads' row blocks
    rng = np.random.default_rng(12)
    n = 10 ** 12 + 7
    vals = np.concatenate([rng.integers(0, n, 4000), 10 ** rng.integers(0, 12, 4000) * rng.integers(1, 1000, 4000),
                           np.array([0, 1, 2, 9, 10, 99999, 100000, 100001, 9999994, 9999995, 9999996, n // 2, n - 1, n, 10 ** 7, 10 ** 7 + 3, 12345649999, 12345650000])])
    K = np.zeros((S, S), np.uint64)
    iu = np.tril_indices(S, -1)
    K[iu] = rng.choice(vals, len(iu[0])).astype(np.uint64)
    K = K + K.T
    text = kg.kinship_format(K, n)
    assert text == ob.kinship_text(K, n)
    cells = text.decode().split("\n")[1].split("\t")
    assert cells[1] == "1" and cells[0] == "%g" % (int(K[1, 0]) / n)


def test_synth_host_twin_matches_numpy_statement():
    for n_acc in (1, 63, 64, 65, 241, 1024, 1135):
        a = kg.synth_rows_host(999, 500, n_acc, 20240601)
        assert (a == synth_rows_numpy(999, 500, n_acc, 20240601)).all()
        assert (a[:, 0] == np.arange(1000, 1500)).all()
    a = kg.synth_rows_host(0, 100, 70, 1)
    b = kg.synth_rows_host(40, 60, 70, 1)
    assert (a[40:] == b).all()  # any shard can be generated independently


def test_compute_fails_loudly_without_a_gpu(have_gpu):
    if have_gpu:
        pytest.skip("a HIP device is present")
    with pytest.raises(kg.KgwasError) as e:
        kg.AssociationScan(64, np.arange(64), np.zeros((1, 64), np.float32), 10, 3)
    assert e.value.code == capi.KGWAS_ERR_DEVICE and "no CPU fallback" in e.value.msg
    with pytest.raises(kg.KgwasError) as e:
        kg.Kinship(64, 3)
    assert e.value.code == capi.KGWAS_ERR_DEVICE


def _run(args):
    return subprocess.run(args, capture_output=True, text=True)


def test_cli_argument_handling(tmp_path):
    a = os.path.join(BIN, "associate_kmers")
    e = os.path.join(BIN, "emma_kinship_kmers")
    r = _run([a, "--help"])
    assert r.returncode == 0 and "--kmers_table" in r.stderr and "--first_phenotype_best" in r.stderr
    r = _run([a, "-p", "x", "-b", "y", "--kmers_table", "t", "--kmer_len", "9"])
    assert r.returncode == 1 and "kmer length has to be between 10-31" in r.stderr
    r = _run([a, "-p", "x"])
    assert r.returncode == 1 and "error parsing options" in r.stderr
    r = _run([a, "--nonsense"])
    assert r.returncode == 1 and "error parsing options" in r.stderr
    r = _run([e, "-t", "x", "-k", "31"])
    assert r.returncode == 1 and "maf is a required parameter" in r.stderr
    r = _run([e, "-t", str(tmp_path / "nope"), "-k", "31", "--maf", "0.05"])
    assert r.returncode == 1 and "Couldn't find file: " in r.stderr and r.stdout == ""
    base, names, rows = _write_table(tmp_path)
    r = _run([e, "-t", base, "-k", "40", "--maf", "0.05"])
    assert r.returncode == 1 and "kmer length has to be between 10-31" in r.stderr
    # data errors abort like the reference's uncaught std::logic_error
    r = _run([e, "-t", base, "-k", "25", "--maf", "0.05"])
    assert r.returncode != 0 and "Kmer length not as defined in class" in r.stderr


def test_cli_fails_loudly_without_a_gpu(tmp_path, have_gpu):
    if have_gpu:
        pytest.skip("a HIP device is present")
    base, names, rows = _write_table(tmp_path)
    ph = tmp_path / "p.tsv"
    ph.write_text("accession_id\tphenotype_value\n" + "".join("%s\t%d\n" % (n, i % 3) for i, n in enumerate(names)))
    r = _run([os.path.join(BIN, "associate_kmers"), "-p", str(ph), "-b", "out", "-o", str(tmp_path), "--kmers_table", base,
              "--kmer_len", "31"])
    assert r.returncode == 3 and "no HIP device" in r.stderr
    assert "Effective minor allele count:\t5" in r.stderr


def _cpu_quota():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        pass
    return n


def test_cli_parallel_is_the_users_cap_and_auto_is_opt_in(tmp_path):
    """src/associate_kmers.cpp:66 honours --parallel; so does the drop-in (round 5 silently raised the pipeline's 1 to every CPU
    of the cgroup). `--parallel 0` / KGWAS_AUTO_PARALLEL=1 opt into the CPU quota. The decision is printed before any device
    work, so this runs with or without a GPU (without one the tool then stops with its no-device error)."""
    base, names, rows = _write_table(tmp_path)
    ph = tmp_path / "p.tsv"
    ph.write_text("accession_id\tphenotype_value\n" + "".join("%s\t%d\n" % (n, i % 3) for i, n in enumerate(names)))
    q = _cpu_quota()
    exe = os.path.join(BIN, "associate_kmers")
    common = ["-p", str(ph), "-b", "out", "-o", str(tmp_path), "--kmers_table", base, "--kmer_len", "31"]

    def requested(extra, env=None):
        e = dict(os.environ)
        e.pop("KGWAS_AUTO_PARALLEL", None)
        e.update(env or {})
        r = subprocess.run([exe] + common + extra, capture_output=True, text=True, env=e)
        m = re.search(r"replay threads requested: (\d+)", r.stderr)
        assert m, r.stderr[-1500:]
        return int(m.group(1)), r.stderr

    n, err = requested(["--parallel", "1"])
    assert n == 1
    assert ("is below the %d CPUs" % q in err) == (q > 1)
    n, err = requested([])  # the reference's default: 4
    assert n == 4
    n, err = requested(["--parallel", "3"])
    assert n == 3
    n, err = requested(["--parallel", "0"])
    assert n == q and "--parallel 0: %d replay threads" % q in err
    n, err = requested(["--parallel", "1"], {"KGWAS_AUTO_PARALLEL": "1"})
    assert n == q and "KGWAS_AUTO_PARALLEL" in err


def test_heap_emulation_is_checked_against_this_process_std_priority_queue():
    """csrc/heap_guard.cpp: once per process the hand emulation of libstdc++'s heap moves is held against a literal
    std::priority_queue over the reference's tuple and comparator (src/kmer_general.h:113-128) on tie / NaN / negative / inf
    streams, single pushes and the lockstep form; every entry point that builds heaps refuses to run on a mismatch
    (KGWAS_ERR_STATE). Flag 1 holds the emulation against a reference with a DIFFERENT tie rule: the check must notice."""
    assert capi.lib.kgwas_heap_selfcheck(0) == capi.KGWAS_OK
    assert capi.lib.kgwas_heap_selfcheck(1) == capi.KGWAS_ERR_STATE
    assert b"differs" in capi.lib.kgwas_last_error()
    assert capi.lib.kgwas_heap_selfcheck(0) == capi.KGWAS_OK  # (the altered run leaves the process's verdict alone)
    h = kg.BestAssociationsHeap(5)  # kgwas_heap_new passes through the guard
    h.add_associations(np.arange(3, dtype=np.uint64), np.array([1.0, 1.0, 2.0]), np.arange(3, dtype=np.uint64))
    assert len(h) == 3


def test_bench_live_traffic_measurement_falls_back_instead_of_failing(monkeypatch):
    """bench.py measures roofline.traffic through rocprofv3 around child runs of itself; whatever goes wrong there - no
    profiler, a child that hangs - must come back as an `error` entry (the line then quotes the committed profile), never as an
    exception or a process left behind."""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.live_pmc_traffic("mx_kernel", 2048 * 512, 8388608, rows=1000, timeout_s=0.05)  # (killed with its process group)
    assert set(r) == {"error"} and "killed" in r["error"]
    monkeypatch.setenv("PATH", "/nonexistent")
    r = bench.live_pmc_traffic("mx_kernel", 2048 * 512, 8388608, rows=1000, timeout_s=5)
    assert set(r) == {"error"} and "rocprofv3" in r["error"]


def test_bench_compact_line_keeps_the_contract_and_fits_the_drivers_tail():
    """bench.py prints ONE JSON line; the driver parses its standard fields and keeps only the last ~9 KB of stdout. The default
    line is the compact form of the full record: every contract field, `roofline` and `cpu_baseline` with their required keys, and
    every judged sub-record - the ones the round-5 verdict could not read LAST - in under 9 KB."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")))
    c = bench.compact_line(full)
    text = json.dumps(c)
    assert len(text) < 9000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in c, k
    assert c["value"] == pytest.approx(full["value"], rel=1e-4) and c["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-4)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    keys = list(c)
    for k in ("north_star_shard", "p1_scan_large", "configs_2_and_4_at_scale", "starved_host", "tie_heavy", "default_topn", "p1_scan", "kinship"):
        assert k in c and keys.index(k) > keys.index("cpu_baseline"), k
    assert keys[-2] == "north_star_shard"
    assert "frac" in c["north_star_shard"]["roofline"] and "host_threads_2" in c["north_star_shard"]
