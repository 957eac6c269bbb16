"""feed / finish split of a configs[1] step (diagnostics)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, M, perms = 1024, 100_000_000, 100
W = 1 + S // 64
Y = make_phenotypes(S, perms, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
for i in range(8):
    t0 = time.perf_counter(); scan.reset(); t1 = time.perf_counter()
    if os.environ.get("EXPECT_FINISH", "1") != "0":
        scan.expect_finish()
    scan.feed_device(table.data_ptr(), M, 0, stream); t2 = time.perf_counter()
    scan.finish(); t3 = time.perf_counter()
    print("reset %.2f ms  feed %.2f ms  finish %.2f ms  total %.2f  popped ahead %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, scan.stats()["columns_popped_ahead"]))
