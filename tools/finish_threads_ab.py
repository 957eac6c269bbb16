"""What rank 0 of an 8-rank run pays at the end of the merge: the north-star shard (250 M x 2048 x 201) scanned with two replay
threads and no finish hint (the merge finishes the session), kgwas_scan_finish timed with KGWAS_FINISH_THREADS = 0 and 16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, P, M = 2048, 201, int(os.environ.get("ROWS", "250000000"))
W = 1 + S // 64
Y = make_phenotypes(S, P - 1, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
for ft in ("0", "16", "0", "16"):
    os.environ["KGWAS_FINISH_THREADS"] = ft
    scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0, host_threads=2)
    for it in range(2):
        scan.reset()
        t0 = time.perf_counter()
        scan.feed_device(table.data_ptr(), M, 0, stream)
        t1 = time.perf_counter()
        scan.finish()
        t2 = time.perf_counter()
    st = scan.stats()
    print("finish threads %2s: feed %.1f ms, finish %.1f ms, total %.1f | selected %d replayed at finish %d" % (ft, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, st["columns_selected"], st["columns_replayed_at_finish"]))
    scan.close()
