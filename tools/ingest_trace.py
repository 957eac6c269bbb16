"""KGWAS_TRACE timeline of a streamed host feed (diagnostics): python tools/ingest_trace.py [rows]"""
import os, sys, time
os.environ["KGWAS_TRACE"] = "1"
os.environ["KGWAS_INGEST_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, P = 1024, 101
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
W = 1 + S // 64
Y = make_phenotypes(S, P - 1, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
host = table.cpu().numpy().view(np.uint64)
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
for it in range(2):
    scan.reset()
    sys.stderr.write("==== pass %d\n" % it); sys.stderr.flush()
    t0 = time.perf_counter()
    scan.feed_host(host, 0); scan.finish()
    sys.stderr.write("==== pass %d: %.1f ms\n" % (it, (time.perf_counter() - t0) * 1e3)); sys.stderr.flush()
