"""Wall time of the two "next" tools' library calls at a realistic size (diagnostics, no checks):
kmers_table_to_bed over a 4 M-row x 1135 table (1.1 GB of .bed), associate_snps over 2 M SNPs x 1135 samples x 101 columns."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmersgwas_amd as kg
from bench import make_phenotypes
S = 1135
d = tempfile.mkdtemp(dir="/tmp")
try:
    rows = 4_000_000
    base = os.path.join(d, "t")
    hdr = np.zeros(16, np.uint8)
    hdr[:4] = np.frombuffer(np.uint32(0xDDCCBBAA).tobytes(), np.uint8)
    hdr[4:12] = np.frombuffer(np.uint64(S).tobytes(), np.uint8)
    hdr[12:16] = np.frombuffer(np.uint32(31).tobytes(), np.uint8)
    with open(base + ".table", "wb") as f:
        f.write(hdr.tobytes())
        for r0 in range(0, rows, 1_000_000):
            kg.synth_rows_host(r0, 1_000_000, S, 20240601).tofile(f)
    names = ["s%d" % i for i in range(S)]
    open(base + ".names", "w").write("".join(n + "\n" for n in names))
    Y = make_phenotypes(S, 100, 7)
    tbl = kg.KmersTable(base, 31)
    for rep in range(2):
        out = os.path.join(d, "bed%d" % rep); os.mkdir(out)
        t0 = time.perf_counter()
        nb, nw = kg.table_to_bed(os.path.join(out, "x"), tbl, np.arange(S, dtype=np.uint64), names, Y[0], kg.min_count(S, 0.05, 5), 10_000_000, False)
        dt = time.perf_counter() - t0
        sz = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
        print("table_to_bed: %d rows -> %d batches, %d k-mers, %.0f MB of files in %.3f s (%.2f GB/s of table, %.2f GB/s written)" % (rows, nb, nw, sz / 1e6, dt, rows * 152 / dt / 1e9, sz / dt / 1e9), flush=True)
        shutil.rmtree(out)
    tbl.close()
    # SNPs
    n_snps = 2_000_000
    rng = np.random.default_rng(3)
    bps = (S + 3) // 4
    sb = os.path.join(d, "snps")
    with open(sb + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        for _ in range(8):
            f.write(rng.integers(0, 256, size=(n_snps // 8) * bps, dtype=np.uint8).tobytes())
    with open(sb + ".bim", "w") as f:
        f.write("".join("1\tsnp%d\t0\t%d\tA\tG\n" % (i, 100 + i) for i in range(n_snps)))
    with open(sb + ".fam", "w") as f:
        f.write("".join("%s %s 0 0 0 -9\n" % (n, n) for n in names))
    t0 = time.perf_counter()
    db = kg.SnpsDataBase(sb, names)
    t1 = time.perf_counter()
    for rep in range(2):
        t2 = time.perf_counter()
        res = db.best(Y, 10001, 57.0)
        t3 = time.perf_counter()
        print("associate_snps: open %.3f s; best of %d SNPs x %d samples x %d columns %.3f s (%.2e SNP x phenotype / s)" % (t1 - t0, n_snps, S, Y.shape[0], t3 - t2, n_snps * Y.shape[0] / (t3 - t2)), flush=True)
    out = os.path.join(d, "snp_out"); os.mkdir(out)
    t0 = time.perf_counter()
    db.write([os.path.join(out, "o.%d" % j) for j in range(Y.shape[0])], res)
    dt = time.perf_counter() - t0
    sz = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print("associate_snps: %d x %d winners written (%.0f MB) in %.3f s" % (Y.shape[0], 10001, sz / 1e6, dt), flush=True)
    db.close()
finally:
    shutil.rmtree(d, ignore_errors=True)
