"""PCIe- and file-inclusive rates of the association scan (never bench.py's `value`): the same synthetic table as
bench.py, (1) resident in HBM, (2) in host memory through kgwas_scan_feed_host, (3) in a .table file (page cache hot)
through kgwas_scan_feed_table. 1024 samples x 101 columns, top-10001."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmersgwas_amd as kg
from bench import make_phenotypes

S, P = 1024, 101
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
W = 1 + S // 64
Y = make_phenotypes(S, P - 1, 7)
mac = kg.min_count(S, 0.05, 5)
col = np.arange(S, dtype=np.uint64)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
host = table.cpu().numpy().view(np.uint64)
d = tempfile.mkdtemp(dir="/tmp")
base = os.path.join(d, "t")
with open(base + ".table", "wb") as f:
    f.write(np.array([0xDDCCBBAA], np.uint32).tobytes() + np.array([0], np.uint32).tobytes()[:0])
    f.seek(0)
    hdr = np.zeros(16, np.uint8)
    hdr[:4] = np.frombuffer(np.uint32(0xDDCCBBAA).tobytes(), np.uint8)
    hdr[4:12] = np.frombuffer(np.uint64(S).tobytes(), np.uint8)
    hdr[12:16] = np.frombuffer(np.uint32(31).tobytes(), np.uint8)
    f.write(hdr.tobytes())
    f.write(host.tobytes())
open(base + ".names", "w").write("".join("s%d\n" % i for i in range(S)))
tbl = kg.KmersTable(base, 31)
scan = kg.AssociationScan(S, col, Y, 10001, mac, device=0)
ref = None
for name, fn in [("HBM-resident  (kgwas_scan_feed_device)", lambda: scan.feed_device(table.data_ptr(), M, 0, stream)),
                 ("host memory   (kgwas_scan_feed_host)  ", lambda: scan.feed_host(host, 0)),
                 (".table file   (kgwas_scan_feed_table) ", lambda: scan.feed_table(tbl, 0, M))]:
    best = 1e9
    for it in range(3):
        scan.reset()
        t0 = time.perf_counter()
        fn()
        scan.finish()
        best = min(best, time.perf_counter() - t0)
    k, sc, r = scan.result(0)
    if ref is None:
        ref = (k.copy(), sc.copy(), r.copy())
    assert (k == ref[0]).all() and sc.tobytes() == ref[1].tobytes() and (r == ref[2]).all()
    print("%s %8.1f ms  %6.2f G rows/s  %6.1f GB/s of table  %.2e kmer*pheno/s" %
          (name, best * 1e3, M / best / 1e9, M * 8 * W / best / 1e9, M * P / best))
os.remove(base + ".table"); os.remove(base + ".names"); os.rmdir(d)
