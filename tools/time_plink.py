"""Times kgwas_write_plink_many (pass 2 of the command-line tool) on a synthetic table in the page cache:
   python tools/time_plink.py [rows=2000000] [S=1135] [P=101] [N=10001]   (CPU only)"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmersgwas_amd as kg
rows, S, P, N = [int(a) for a in sys.argv[1:5]] + [2_000_000, 1135, 101, 10001][len(sys.argv) - 1:]
d = tempfile.mkdtemp(dir=os.environ.get("KGWAS_BENCH_TMP", "/tmp"))
try:
    W = 1 + (S + 63) // 64
    base = os.path.join(d, "t")
    hdr = np.zeros(16, np.uint8)
    hdr[:4] = np.frombuffer(np.uint32(0xDDCCBBAA).tobytes(), np.uint8)
    hdr[4:12] = np.frombuffer(np.uint64(S).tobytes(), np.uint8)
    hdr[12:16] = np.frombuffer(np.uint32(31).tobytes(), np.uint8)
    with open(base + ".table", "wb") as f:
        f.write(hdr.tobytes())
        step = 500_000
        for r0 in range(0, rows, step):
            kg.synth_rows_host(r0, min(step, rows - r0), S, 1).tofile(f)
    names = ["s%d" % i for i in range(S)]
    open(base + ".names", "w").write("".join(n + "\n" for n in names))
    t = kg.KmersTable(base, 31)
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((P, S)).astype(np.float32)
    picks = [rng.choice(rows, size=N, replace=False).astype(np.uint64) for _ in range(P)]
    kmers = [p + 1 for p in picks]
    outs = [os.path.join(d, "o%d" % j) for j in range(P)]
    col = np.arange(S, dtype=np.uint64)
    for rep in range(3):
        t0 = time.perf_counter()
        kg.write_plink_many(outs, t, col, names, Y, kmers, picks)
        t1 = time.perf_counter()
        print("write_plink_many: %d columns x %d winners, %d samples: %.3f s" % (P, N, S, t1 - t0))
    t0 = time.perf_counter()
    for j in range(3):
        kg.write_plink(outs[j], t, col, names, Y[j], kmers[j], picks[j])
    print("one column at a time: %.4f s per column" % ((time.perf_counter() - t0) / 3))
finally:
    shutil.rmtree(d, ignore_errors=True)
