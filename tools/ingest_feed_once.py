"""One streamed host feed (for profilers): python tools/ingest_feed_once.py [rows] [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, P = 1024, 101
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W = 1 + S // 64
Y = make_phenotypes(S, P - 1, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
host = table.cpu().numpy().view(np.uint64)
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
for it in range(passes):
    scan.reset()
    t0 = time.perf_counter()
    scan.feed_host(host, 0); scan.finish()
    print("pass %d: %.1f ms = %.1f GB/s" % (it, (time.perf_counter() - t0) * 1e3, M * W * 8 / (time.perf_counter() - t0) / 1e9), flush=True)
