"""GPU tests of the two drop-in command-line tools at the process boundary (argv + files in, files out),
byte-for-byte against what the oracle writes for the same inputs."""
import os
import subprocess

import numpy as np
import pytest

import kmersgwas_amd as kg
from oracle import binding as ob
from oracle import oracle_np as onp
from helpers import random_table, phenotypes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kmersgwas_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden")


def _oracle_outputs(outdir, base, rows, S_f, names, acc, pnames, Y, topn, mac, k, scores=False):
    col = onp.column_map(names, acc)
    res = ob.associate(rows, S_f, col, Y, topn, mac)
    for j, pn in enumerate(pnames):
        o = res["per_pheno"][j]
        ob.write_plink(os.path.join(outdir, "%s.%d.%s" % (base, j, pn)), rows, S_f, col, acc, Y[j], k, o["kmer"], o["file_row"])
        if scores:
            with open(os.path.join(outdir, "%s.%d.best_kmers.scores" % (base, j)), "wb") as f:
                for km, sc in zip(o["kmer"], o["score"]):
                    f.write(np.uint64(km).tobytes() + np.float64(sc).tobytes())
    open(os.path.join(outdir, base + ".tested_kmers"), "w").write("%d\n" % res["tested"])
    return res


def _compare_dirs(a, b):
    fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
    assert fa == fb, (fa, fb)
    for f in fa:
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f
    return fa


def test_associate_kmers_ecoli_shaped_config(tmp_path):
    """BASELINE.json configs[0]: the E. coli example's phenotype (241 accessions, binary) on a synthetic
    241-sample table with many duplicated presence/absence patterns, pipeline-style invocation."""
    names, acc, Y = onp.load_phenotypes(os.path.join(GOLD, "resistence.pheno"))
    S_f, k = 241, 31
    table_names = list(reversed(acc))  # table column order differs from the phenotype file's order
    rows = random_table(200_000, S_f, seed=2024, dup_frac=0.5)
    base = str(tmp_path / "kmers_table")
    onp.write_table(base, table_names, k, rows[:, 0], rows[:, 1:])
    out_p, out_o = tmp_path / "prod", tmp_path / "orc"
    out_p.mkdir(); out_o.mkdir()
    cmd = [os.path.join(BIN, "associate_kmers"), "-p", os.path.join(GOLD, "resistence.pheno"), "-b", "pheno", "-o", str(out_p),
           "-n", "10001", "--parallel", "4", "--kmers_table", base, "--kmer_len", "31", "--maf", "0.050000", "--mac", "5",
           "--pattern_counter"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Effective minor allele count:\t13" in r.stderr
    mac = onp.min_count(241, 0.05, 5)
    res = _oracle_outputs(str(out_o), "pheno", rows, S_f, table_names, acc, names, Y, 10001, mac, k)
    pat = ob.associate(rows, S_f, onp.column_map(table_names, acc), Y, 10001, mac, count_patterns=True)["patterns"]
    open(os.path.join(str(out_o), "pheno.pattern_counter"), "w").write("%d\n" % pat)  # src/associate_kmers.cpp:197-201
    assert "Total patterns\t%d" % pat in r.stderr
    files = _compare_dirs(str(out_p), str(out_o))
    assert files == ["pheno.0.phenotype_value.bed", "pheno.0.phenotype_value.bim", "pheno.0.phenotype_value.fam",
                     "pheno.pattern_counter", "pheno.tested_kmers"]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"KGWAS_PIN_PLAIN": "1", "KGWAS_CLI_FULL_TEARDOWN": "1"}, {"KGWAS_RECORD_COPY": "memcpy"},
                                 {"KGWAS_RECORD_COPY": "kernel", "KGWAS_INGEST_DEVICE": "2", "KGWAS_INGEST_PINNED": "2", "KGWAS_INGEST_PIECE_ROWS": "4096"}])
def test_associate_kmers_process_switches_change_nothing(tmp_path, env):
    """The start-up / teardown / transfer switches of the tool (hipHostMalloc instead of registered huge-page mappings, an orderly
    teardown instead of _exit, the records' two ways to the host, a minimal ingest ring with tiny pieces):
    the same files as the default run, byte for byte."""
    names, acc, Y = onp.load_phenotypes(os.path.join(GOLD, "resistence.pheno"))
    S_f = 241
    rows = random_table(120_000, S_f, seed=77, dup_frac=0.3)
    base = str(tmp_path / "kmers_table")
    onp.write_table(base, list(acc), 31, rows[:, 0], rows[:, 1:])
    outs = []
    for tag, e in (("default", {}), ("switched", env)):
        out = tmp_path / tag
        out.mkdir()
        cmd = [os.path.join(BIN, "associate_kmers"), "-p", os.path.join(GOLD, "resistence.pheno"), "-b", "pheno", "-o", str(out), "-n", "1001",
               "--kmers_table", base, "--kmer_len", "31", "--maf", "0.050000", "--mac", "5"]
        r = subprocess.run(cmd, capture_output=True, text=True, env={**os.environ, **e})
        assert r.returncode == 0, r.stderr[-2000:]
        if e.get("KGWAS_CLI_FULL_TEARDOWN"):
            assert "[kgwas] teardown_s=" in r.stderr
        outs.append(str(out))
    assert len(_compare_dirs(outs[0], outs[1])) == 4


@pytest.mark.parametrize("kernel", ["1", "2", "3"])
def test_associate_kmers_permutations_subset_scores(tmp_path, kernel):
    """Several phenotype columns (value + permutations), phenotyped subset in shuffled order,
    --first_phenotype_best, --k_mers_scores, small --batch_size: every output file byte-identical."""
    S_f, S, k, P = 300, 257, 25, 6
    rows = random_table(30_000, S_f, seed=9, dup_frac=0.2)
    table_names = ["acc_%d" % i for i in range(S_f)]
    base = str(tmp_path / "tab")
    onp.write_table(base, table_names, k, rows[:, 0] & np.uint64((1 << 50) - 1), rows[:, 1:])
    rows[:, 0] &= np.uint64((1 << 50) - 1)
    pick = np.random.default_rng(3).permutation(S_f)[:S]
    acc = [table_names[i] for i in pick]
    Y = phenotypes(S, P - 1, seed=21)
    pnames = ["phenotype_value"] + ["P%d" % i for i in range(1, P)]
    ph = tmp_path / "ph.tsv"
    with open(ph, "w") as f:
        f.write("accession_id\t" + "\t".join(pnames) + "\n")
        for i, a in enumerate(acc):
            f.write(a + "\t" + "\t".join(repr(float(Y[j, i])) for j in range(P)) + "\n")
    names2, acc2, Y2 = onp.load_phenotypes(str(ph))
    assert Y2.tobytes() == Y.tobytes()
    out_p, out_o = tmp_path / "prod", tmp_path / "orc"
    out_p.mkdir(); out_o.mkdir()
    cmd = [os.path.join(BIN, "associate_kmers"), "--phenotype_file", str(ph), "--base_name", "x", "--output_dir", str(out_p),
           "--best", "500", "--first_phenotype_best", "1500", "--kmers_table", base, "--kmer_len", "25", "--maf=0.05",
           "--mac", "5", "--k_mers_scores", "--batch_size", "7001", "--kernel", kernel]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    topn = np.full(P, 500, np.uint64)
    topn[0] = 1500
    _oracle_outputs(str(out_o), "x", rows, S_f, table_names, acc, pnames, Y, topn, onp.min_count(S, 0.05, 5), k, scores=True)
    files = _compare_dirs(str(out_p), str(out_o))
    assert len(files) == 3 * P + P + 1
    # unknown accession -> the reference's uncaught logic_error (abort)
    with open(ph, "a") as f:
        f.write("not_in_table\t" + "\t".join(["1"] * P) + "\n")
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and "Couldn't find path for DB: not_in_table" in r.stderr


@pytest.mark.parametrize("gpus", ["2", "3"])
def test_associate_kmers_gpus_row_sharded(tmp_path, gpus):
    """--gpus N: the table row-sharded over N scan sessions inside the one process (kgwas_multiscan; with fewer GPUs
    present the shards share them), later shards merged into the first: every output file byte-identical to the
    oracle's single scan, --pattern_counter and .tested_kmers included. Heavy ties (binary trait, duplicated rows)."""
    names, acc, Y = onp.load_phenotypes(os.path.join(GOLD, "resistence.pheno"))
    S_f, k = 241, 31
    table_names = list(reversed(acc))
    rows = random_table(150_001, S_f, seed=606, dup_frac=0.5)
    base = str(tmp_path / "kmers_table")
    onp.write_table(base, table_names, k, rows[:, 0], rows[:, 1:])
    out_p, out_o = tmp_path / "prod", tmp_path / "orc"
    out_p.mkdir(); out_o.mkdir()
    cmd = [os.path.join(BIN, "associate_kmers"), "-p", os.path.join(GOLD, "resistence.pheno"), "-b", "pheno", "-o", str(out_p),
           "-n", "2001", "--parallel", "4", "--kmers_table", base, "--kmer_len", "31", "--maf", "0.050000", "--mac", "5",
           "--pattern_counter", "--gpus", gpus]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "gpus=%s" % gpus in r.stderr
    mac = onp.min_count(241, 0.05, 5)
    _oracle_outputs(str(out_o), "pheno", rows, S_f, table_names, acc, names, Y, 2001, mac, k)
    pat = ob.associate(rows, S_f, onp.column_map(table_names, acc), Y, 2001, mac, count_patterns=True)["patterns"]
    open(os.path.join(str(out_o), "pheno.pattern_counter"), "w").write("%d\n" % pat)
    _compare_dirs(str(out_p), str(out_o))


def _quota():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        pass
    return n


def test_associate_kmers_parallel_is_honoured_and_auto_is_opt_in(tmp_path):
    """--parallel is the user's cap, as in the reference (src/associate_kmers.cpp:66): kmers_gwas.py's default 1
    (src/py/pipeline_parser.py:31) means ONE replay thread (per GPU at least one), with a hint on stderr when the process may
    use more CPUs; `--parallel 0` and KGWAS_AUTO_PARALLEL=1 take the CPU quota (divided among the GPUs). Results do not
    depend on it."""
    import re
    names, acc, Y = onp.load_phenotypes(os.path.join(GOLD, "resistence.pheno"))
    S_f, k = 241, 31
    rows = random_table(30_001, S_f, seed=91, dup_frac=0.3)
    base = str(tmp_path / "kmers_table")
    onp.write_table(base, acc, k, rows[:, 0], rows[:, 1:])
    outs = []
    q = _quota()
    for name, par, auto_env, want in (("one", "1", False, 1), ("zero", "0", False, q), ("env", "1", True, q), ("three", "3", False, 3)):
        out = tmp_path / name
        out.mkdir()
        cmd = [os.path.join(BIN, "associate_kmers"), "-p", os.path.join(GOLD, "resistence.pheno"), "-b", "pheno", "-o", str(out),
               "-n", "501", "--parallel", par, "--kmers_table", base, "--kmer_len", "31", "--maf", "0.050000", "--mac", "5", "--gpus", "2"]
        env = dict(os.environ)
        env.pop("KGWAS_AUTO_PARALLEL", None)
        if auto_env:
            env["KGWAS_AUTO_PARALLEL"] = "1"
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        m = re.search(r"replay_threads=(\d+) replay_threads_per_gpu=(\d+)", r.stderr)
        assert m, r.stderr[-2000:]
        assert (int(m.group(1)), int(m.group(2))) == (want, max(1, want // 2)), r.stderr[-2000:]
        if name in ("one", "three") and want < q:
            assert "is below the %d CPUs" % q in r.stderr and "keeps to %s thread(s)" % par in r.stderr
        outs.append(str(out))
    for o in outs[1:]:
        _compare_dirs(outs[0], o)


@pytest.mark.parametrize("ranks,merge,shape", [(2, "root", "small"), (3, "column", "small"), (2, "root", "north_star"), (2, "column", "north_star")])
def test_bench_ranks_merge_over_torch_distributed(ranks, merge, shape):
    """bench.py's N > 1 path with real scan sessions: `ranks` processes (torch.distributed over gloo, all on the one GPU of
    the test box), each scanning its own row shard, merged with kmersgwas_amd.dist (to the root / by column); rank 0 then
    scans all the rows in one session and compares every column's heap bytes (--check-merge)."""
    import json, sys
    env = dict(os.environ, KGWAS_DIST_BACKEND="gloo", KGWAS_BENCH_MERGE=merge, KGWAS_PIN_THREADS="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + ranks), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "1",
           "--rows", "600000", "--samples", "241", "--perms", "12", "--topn", "2001", "--check-merge"]
    if shape == "north_star":  # BASELINE configs[3]'s columns and heap size: 2048 samples x 201 columns, top-10001, 2 M rows per rank
        cmd[cmd.index("--rows") + 1:cmd.index("--check-merge")] = ["2000000", "--samples", "2048", "--perms", "200", "--topn", "10001"]
        cmd[cmd.index("--master-port") + 1] = str(29520 + ranks + (1 if merge == "column" else 0))
    if ranks != 2 or shape == "north_star":
        cmd.append("--no-cpu-baseline")  # (with two ranks the line also carries rank 0's shard-parity check against the oracle)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == ranks and j["merge_check"] is True
    assert j["single_gpu_same_shard"]["value"] > 0
    # every rank's own record: which rank, and which side of it (GPU kernels, host replay, merge), set the pace
    assert [x["rank"] for x in j["ranks"]] == list(range(ranks))
    for x in j["ranks"]:
        # (a rank's columns are replayed - pushes - or kept in select mode and never touch a heap, scan_lazy.cpp)
        assert x["step_ms"] > 0 and x["kernels_ms"] > 0 and x["replay_threads"] >= 1 and x["candidates"] > 0
    if ranks == 2 and shape == "small":
        assert j["parity_check"] is True


def test_emma_kinship_kmers_gpus(tmp_path):
    S_f, k = 173, 31
    rows = random_table(40_001, S_f, seed=78)
    names = ["s%d" % i for i in range(S_f)]
    base = str(tmp_path / "tab")
    onp.write_table(base, names, k, rows[:, 0], rows[:, 1:])
    r = subprocess.run([os.path.join(BIN, "emma_kinship_kmers"), "-t", base, "-k", "31", "--maf", "0.05", "--gpus", "3"], capture_output=True)
    assert r.returncode == 0, r.stderr[-2000:]
    mc = int(np.ceil(S_f * 0.05))
    K, n = ob.kinship(rows, S_f, mc)
    assert r.stdout == ob.kinship_text(K, n)


def test_emma_kinship_kmers_stdout(tmp_path):
    S_f, k = 241, 31
    rows = random_table(50_000, S_f, seed=77)
    names = ["s%d" % i for i in range(S_f)]
    base = str(tmp_path / "tab")
    onp.write_table(base, names, k, rows[:, 0], rows[:, 1:])
    r = subprocess.run([os.path.join(BIN, "emma_kinship_kmers"), "-t", base, "-k", "31", "--maf", "0.05"], capture_output=True)
    assert r.returncode == 0, r.stderr[-2000:]
    mc = int(np.ceil(S_f * 0.05))
    K, n = ob.kinship(rows, S_f, mc)
    assert r.stdout == ob.kinship_text(K, n)
    err = r.stderr.decode()
    assert "Min count = %d" % mc in err and "#%d" % n in err


@pytest.mark.parametrize("unique", [False, True])
@pytest.mark.parametrize("S_f,n_pick,batch", [(241, 241, 1000), (300, 257, 777), (70, 33, 100000)])
def test_kmers_table_to_bed(tmp_path, unique, S_f, n_pick, batch):
    """the table -> PLINK export tool. Batches count KEPT k-mers, a batch exists iff rows
    were left when it started, -u keeps the first k-mer of each presence/absence hash over all batches: every file
    byte-identical to the oracle's, through the CLI and through the library."""
    k = 31
    rows = random_table(12_345, S_f, seed=S_f + batch, dup_frac=0.4)
    rows[-40:, 1:] = 0  # trailing rows that fail the MAC filter (a last batch with nothing kept can exist)
    table_names = ["acc%d" % i for i in range(S_f)]
    base = str(tmp_path / "tab")
    onp.write_table(base, table_names, k, rows[:, 0], rows[:, 1:])
    pick = np.random.default_rng(5).permutation(S_f)[:n_pick]
    acc = [table_names[i] for i in pick]
    y = phenotypes(n_pick, 0, seed=2)[0]
    ph = tmp_path / "ph.tsv"
    with open(ph, "w") as f:
        f.write("accession_id\tphenotype_value\n")
        for a, v in zip(acc, y):
            f.write("%s\t%r\n" % (a, float(v)))
    names2, acc2, Y2 = onp.load_phenotypes(str(ph))
    col = onp.column_map(table_names, acc2)
    mc = max(int(np.ceil(n_pick * 0.05)), 5)
    out_o, out_p, out_l = tmp_path / "orc", tmp_path / "cli", tmp_path / "lib"
    for d in (out_o, out_p, out_l):
        d.mkdir()
    nb, nw = ob.table_to_bed(str(out_o / "x"), rows, S_f, col, acc2, Y2[0], k, mc, batch, unique)
    assert nb >= 1 and nw > 0
    cmd = [os.path.join(BIN, "kmers_table_to_bed"), "-t", base, "-k", str(k), "-p", str(ph), "--maf", "0.05", "--mac", "5",
           "-b", str(batch), "-o", str(out_p / "x")] + (["-u"] if unique else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "using phenotype_value" in r.stderr and r.stderr.count("Batch:") == nb
    files = _compare_dirs(str(out_p), str(out_o))
    assert len(files) == 3 * nb
    tbl = kg.KmersTable(base, k)
    assert kg.table_to_bed(str(out_l / "x"), tbl, col, acc2, Y2[0], mc, batch, unique) == (nb, nw)
    tbl.close()
    _compare_dirs(str(out_l), str(out_o))
    # option errors behave like the reference's
    r = subprocess.run(cmd[:-2] if not unique else cmd[:-3], capture_output=True, text=True)  # no -o
    assert r.returncode == 1 and "is a required parameter" in r.stderr


def _write_bedbimfam(base, n_samples, n_snps, seed):
    """A random PLINK trio: every dubit value occurs (hom. minor 00, missing 01, het 10, hom. major 11)."""
    rng = np.random.default_rng(seed)
    dub = rng.choice(4, size=(n_snps, n_samples), p=[0.45, 0.05, 0.2, 0.3]).astype(np.uint8)
    dub[:5] = 0            # monomorphic SNPs (fail the MAC test -> score 0, but still offered to the heap)
    dub[5:8] = 1           # all missing: N = 0
    dub[8, : n_samples // 2] = 3  # a duplicate pair of SNPs: equal scores
    dub[9] = dub[8]
    bps = (n_samples + 3) // 4
    body = np.zeros((n_snps, bps), np.uint8)
    for s_ in range(n_samples):
        body[:, s_ // 4] |= (dub[:, s_] << ((s_ % 4) * 2)).astype(np.uint8)
    names = ["smp%d" % i for i in range(n_samples)]
    with open(base + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]) + body.tobytes())
    with open(base + ".bim", "w") as f:
        for i in range(n_snps):
            f.write("1\tsnp%d\t0\t%d\tA\tG\n" % (i, 100 + i))
    with open(base + ".fam", "w") as f:
        for n in names:
            f.write("%s %s 0 0 0 -9\n" % (n, n))
    return names, body


@pytest.mark.parametrize("n_file,n_use,n_snps", [(53, 53, 400), (300, 257, 3000), (1030, 900, 1500)])
def test_associate_snps(tmp_path, n_file, n_use, n_snps):
    """the SNP twin of the scorer. Scores bit-identical to the oracle's restatement of
    calculate_grammmar_approx_association (missing and heterozygous calls, MAC failures = 0, 0/0 = NaN at mac 0), the
    per-phenotype top-N index lists equal a literal std::priority_queue's, and the tool's output files equal the
    lines / bytes those lists select."""
    base = str(tmp_path / "snps")
    names, body = _write_bedbimfam(base, n_file, n_snps, seed=n_file)
    pick = np.random.default_rng(1).permutation(n_file)[:n_use]
    use = [names[i] for i in pick]
    P, topn = 3, 57
    Y = phenotypes(n_use, P - 1, seed=4)
    pnames = ["trait%d" % j for j in range(P)]
    ph = tmp_path / "ph.tsv"
    with open(ph, "w") as f:
        f.write("accession_id\t" + "\t".join(pnames) + "\n")
        for i, a in enumerate(use):
            f.write(a + "\t" + "\t".join(repr(float(Y[j, i])) for j in range(P)) + "\n")
    mac = float(max(np.ceil(0.05 * n_use), 5.0))
    exp = np.stack([ob.snps_scores(body.tobytes(), n_file, pick, Y[j], mac) for j in range(P)])
    db = kg.SnpsDataBase(base, use)
    assert (db.n_snps, db.n_samples_file) == (n_snps, n_file)
    got = db.scores(Y, mac)
    assert got.tobytes() == exp.tobytes()
    assert (exp[:, :8] == 0).all() and (exp[:, 8] == exp[:, 9]).all() and (exp[:, 10:] > 0).any()
    got0 = db.scores(Y[:1], 0.0)  # mac = 0: nothing is filtered, all-missing SNPs divide 0 by 0
    exp0 = ob.snps_scores(body.tobytes(), n_file, pick, Y[0], 0.0)
    assert got0[0].tobytes() == exp0.tobytes() and np.isnan(exp0[5:8]).all()
    best = db.best(Y, topn, mac)
    lists = []
    for j in range(P):
        h = ob.Heap(topn)
        h.add_many(np.zeros(n_snps, np.uint64), exp[j], np.arange(n_snps, dtype=np.uint64))
        rows = np.sort(h.pop_all()[2])
        lists.append(rows)
        assert (best[j] == rows).all()
    out_p, out_o = tmp_path / "prod", tmp_path / "orc"
    out_p.mkdir(); out_o.mkdir()
    bim_lines = open(base + ".bim").read().split("\n")
    for j in range(P):  # output_plink_bed_file, restated: the selected .bim lines and .bed rows
        with open(out_o / ("o.%s.bed" % pnames[j]), "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x01]) + body[lists[j].astype(np.int64)].tobytes())
        with open(out_o / ("o.%s.bim" % pnames[j]), "w") as f:
            f.write("".join(bim_lines[int(i)] + "\n" for i in lists[j]))
    r = subprocess.run([os.path.join(BIN, "associate_snps"), str(ph), base, str(out_p / "o"), str(topn), "0.05", "5"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "(snps,samples) = %d, %d" % (n_snps, n_file) in r.stderr and "Minor allele count  = %g" % mac in r.stderr
    _compare_dirs(str(out_p), str(out_o))
    db.close()
    # a phenotyped sample that is not in the .fam: the reference's uncaught logic_error
    with pytest.raises(kg.KgwasError) as e:
        kg.SnpsDataBase(base, use + ["nobody"])
    assert "All accessions should be in fam file: nobody" in str(e.value)
    r = subprocess.run([os.path.join(BIN, "associate_snps"), str(ph)], capture_output=True, text=True)
    assert r.returncode == 1 and "usage:" in r.stderr


def test_associate_kmers_without_n_uses_the_reference_default_of_a_million(tmp_path):
    """No `-n`: the reference's default heap size 1 000 000 (src/associate_kmers.cpp:44) - on a table with more MAC-passing rows
    than that, so the heaps fill and the default is visible in the outputs: 1 000 000 winners per column, files identical to
    the oracle's."""
    S_f, k, P = 64, 31, 2
    rows = kg.synth_rows_host(0, 1_300_000, S_f, 99)
    names = ["a%d" % i for i in range(S_f)]
    base = str(tmp_path / "kmers_table")
    onp.write_table(base, names, k, rows[:, 0], rows[:, 1:])
    Y = phenotypes(S_f, P - 1, seed=5)
    pnames = ["phenotype_value", "P1"]
    ph = tmp_path / "ph.tsv"
    with open(ph, "w") as f:
        f.write("accession_id\t" + "\t".join(pnames) + "\n")
        for i, a in enumerate(names):
            f.write(a + "\t" + "\t".join(repr(float(Y[j, i])) for j in range(P)) + "\n")
    pn, acc, Yl = onp.load_phenotypes(str(ph))
    out_p, out_o = tmp_path / "prod", tmp_path / "orc"
    out_p.mkdir(); out_o.mkdir()
    cmd = [os.path.join(BIN, "associate_kmers"), "-p", str(ph), "-b", "pheno", "-o", str(out_p), "--parallel", "4",
           "--kmers_table", base, "--kmer_len", "31", "--maf", "0.050000", "--mac", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    mac = onp.min_count(S_f, 0.05, 5)
    res = _oracle_outputs(str(out_o), "pheno", rows, S_f, names, acc, pn, Yl, 1_000_000, mac, k)
    assert res["tested"] > 1_000_000 and len(res["per_pheno"][0]["kmer"]) == 1_000_000
    files = _compare_dirs(str(out_p), str(out_o))
    assert len(files) == 3 * P + 1
