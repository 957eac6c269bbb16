import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KGWAS_TRACE"] = "1"
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, P, M = 1024, 101, 100_000_000
W = 1 + S // 64
Y = make_phenotypes(S, P - 1, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
scan.feed_device(table.data_ptr(), M, 0, stream)
scan.reset()
sys.stderr.write("==== second pass\n")
import time
scan.expect_finish()
t0 = time.perf_counter()
scan.feed_device(table.data_ptr(), M, 0, stream)
t1 = time.perf_counter()
scan.finish()
t2 = time.perf_counter()
sys.stderr.write("==== feed %.2f ms, finish %.2f ms; columns selected %d, replayed at finish %d\n" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, scan.stats()["columns_selected"], scan.stats()["columns_replayed_at_finish"]))
