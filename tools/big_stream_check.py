"""One-off consistency check at a size beyond every 32-bit offset: a table of `rows` x 1135 samples (default 400 M rows =
60.8 GB) scanned (a) resident in HBM in one feed, (b) streamed from host memory through the ingest pipeline, (c) streamed
in three ragged feeds - identical heaps, push and tested counts.   python tools/big_stream_check.py [rows] [columns]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = 1135
W = 1 + (S + 63) // 64
Y = make_phenotypes(S, P - 1, 7)
mac = kg.min_count(S, 0.05, 5)
col = np.arange(S, dtype=np.uint64)
table = torch.empty(rows * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, rows, S, 20240601, stream)
torch.cuda.synchronize()
def results(scan):
    st = scan.stats()
    return [tuple(a.tobytes() for a in scan.result(j)) for j in range(P)], st["heap_pushes"], st["rows_tested"]
scan = kg.AssociationScan(S, col, Y, 10001, mac)
t0 = time.perf_counter(); scan.feed_device(table.data_ptr(), rows, 0, stream); scan.finish(); t1 = time.perf_counter()
ref = results(scan)
print("resident: %.1f GB in %.3f s, tested %d pushes %d" % (rows * 8.0 * W / 1e9, t1 - t0, ref[2], ref[1]))
t0 = time.perf_counter(); host = table.cpu().numpy().view(np.uint64).reshape(rows, W); t1 = time.perf_counter()
print("copied to host in %.1f s" % (t1 - t0))
del table; torch.cuda.empty_cache()
scan.reset()
t0 = time.perf_counter(); scan.feed_host(host, 0); scan.finish(); t1 = time.perf_counter()
a = results(scan)
print("streamed from host memory, one feed: %.3f s = %.1f GB/s, identical %s" % (t1 - t0, rows * 8.0 * W / 1e9 / (t1 - t0), a == ref))
scan.reset()
c1, c2 = rows // 3 + 12345, 2 * rows // 3 + 77
t0 = time.perf_counter()
scan.feed_host(host[:c1], 0); scan.feed_host(host[c1:c2], c1); scan.expect_finish(); scan.feed_host(host[c2:], c2); scan.finish()
t1 = time.perf_counter()
b = results(scan)
print("streamed in three ragged feeds: %.3f s, identical %s" % (t1 - t0, b == ref))
assert a == ref and b == ref
