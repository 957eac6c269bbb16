"""Cost of keeping the cross-shard history during a pass: record_history 0 (off), 1 (full log), 2 (eviction ring)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
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
for rh in (0, 1, 2, 0, 1, 2):
    scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0, record_history=rh)
    ts = []
    for it in range(5):
        scan.reset()
        t0 = time.perf_counter()
        scan.feed_device(table.data_ptr(), M, 0, stream)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("record_history=%s: feed ms %s replay %.1f" % (rh, [round(x, 1) for x in ts], scan.stats()["replay_ms"]))
    scan.close()
