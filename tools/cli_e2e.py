"""Runs kmersgwas_amd/bin/associate_kmers end to end on a synthetic .table (page cache) and prints where its wall time went:
   python tools/cli_e2e.py [rows=40000000] [S=1135] [P=101] [N=10001]      (needs a GPU)
KGWAS_TRACE=1 in the environment adds the pass-2 writer's own phases."""
import os, sys, time, tempfile, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmersgwas_amd as kg
from bench import make_phenotypes
rows, S, P, N = [int(a) for a in sys.argv[1:5]] + [40_000_000, 1135, 101, 10001][len(sys.argv) - 1:]
d = tempfile.mkdtemp(dir=os.environ.get("KGWAS_BENCH_TMP", "/tmp"))
try:
    base = os.path.join(d, "t")
    hdr = np.zeros(16, np.uint8)
    hdr[:4] = np.frombuffer(np.uint32(0xDDCCBBAA).tobytes(), np.uint8)
    hdr[4:12] = np.frombuffer(np.uint64(S).tobytes(), np.uint8)
    hdr[12:16] = np.frombuffer(np.uint32(31).tobytes(), np.uint8)
    with open(base + ".table", "wb") as f:
        f.write(hdr.tobytes())
        for r0 in range(0, rows, 1_000_000):
            kg.synth_rows_host(r0, min(1_000_000, rows - r0), S, 20240601).tofile(f)
    if not os.environ.get("NO_SYNC"):
        t0 = time.perf_counter()
        os.sync()  # the table's dirty pages: while they are written back, every file creation waits for the journal
        print("sync after writing the table: %.2f s" % (time.perf_counter() - t0))
    open(base + ".names", "w").write("".join("s%d\n" % i for i in range(S)))
    Y = make_phenotypes(S, P - 1, 7)
    pheno = os.path.join(d, "p.pheno")
    with open(pheno, "w") as f:
        f.write("accession_id\t" + "\t".join("perm%d" % j for j in range(P)) + "\n")
        for i in range(S):
            f.write("s%d\t" % i + "\t".join("%.9g" % float(Y[j, i]) for j in range(P)) + "\n")
    exe = os.path.join(ROOT, "kmersgwas_amd", "bin", "associate_kmers")
    for rep in range(3):
        out = os.path.join(os.environ.get("OUT_ROOT", d), "out%d_%d" % (os.getpid(), rep))
        os.makedirs(out)
        t0 = time.perf_counter()
        wrap = os.environ.get("CLI_WRAP", "").split()  # e.g. a profiler in front of the binary
        r = subprocess.run(wrap + [exe, "-p", pheno, "-b", "run", "-o", out, "-n", str(N), "--parallel", "1", "--kmers_table", base, "--kmer_len", "31",
                            "--maf", "0.050000", "--mac", "5"], capture_output=True, text=True)
        wall = time.perf_counter() - t0
        print("run %d: rc %d wall %.3f s" % (rep, r.returncode, wall))
        for l in r.stderr.splitlines():
            if "seconds:" in l or "teardown" in l or "write_plink_many" in l or "scan_create" in l:
                print("   ", l)
    if os.environ.get("KIN"):  # the kinship tool on the same table
        kexe = os.path.join(ROOT, "kmersgwas_amd", "bin", "emma_kinship_kmers")
        for rep in range(3):
            t0 = time.perf_counter()
            r = subprocess.run([kexe, "-t", base, "-k", "31", "--maf", "0.05"], capture_output=True)
            wall = time.perf_counter() - t0
            print("emma_kinship_kmers run %d: rc %d wall %.3f s (%.1f GB/s of table), stdout %d bytes" % (rep, r.returncode, wall, rows * (1 + (S + 63) // 64) * 8 / wall / 1e9, len(r.stdout)))
            print("   ", [l for l in r.stderr.decode().splitlines() if "seconds:" in l][-1:])
finally:
    shutil.rmtree(d, ignore_errors=True)
