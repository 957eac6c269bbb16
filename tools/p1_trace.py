"""Timeline of a one-column pass (100 M x 1024, top-10001) from the session's KGWAS_TRACE prints: when each chunk was
submitted, its counts arrived, it was published, and when the (single) replay worker finished it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, M = 1024, 100_000_000
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = 1 + S // 64
Y = make_phenotypes(S, max(P - 1, 0), 7)[:P]
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
for _ in range(2):
    scan.reset()
    scan.feed_device(table.data_ptr(), M, 0, stream)
    scan.finish()
os.environ["KGWAS_TRACE"] = "1"
scan2 = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
scan2.feed_device(table.data_ptr(), M, 0, stream)
scan2.finish()
scan2.reset()
sys.stderr.write("==== traced pass\n")
sys.stderr.flush()
import time
t0 = time.perf_counter()
scan2.feed_device(table.data_ptr(), M, 0, stream)
t1 = time.perf_counter()
scan2.finish()
sys.stderr.write("==== feed %.3f ms, finish %.3f ms\n" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
