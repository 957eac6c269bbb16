"""Cost of --pattern_counter: the same 100 M x 1024 x 101 scan with and without the pattern pass (hash + distinct count)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, P = 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 101
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
W = 1 + S // 64
Y = make_phenotypes(S, P - 1, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(rows * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, rows, S, 20240601, stream)
torch.cuda.synchronize()
for cp in (False, True, True):
    scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0, count_patterns=cp)
    ts = []
    for it in range(3):
        scan.reset()
        t0 = time.perf_counter()
        scan.feed_device(table.data_ptr(), rows, 0, stream); scan.finish()
        ts.append((time.perf_counter() - t0) * 1e3)
    st = scan.stats()
    print("count_patterns=%s: %.1f ms per pass (min of %s); patterns %s" % (cp, min(ts), ["%.1f" % t for t in ts], st.get("patterns")), flush=True)
    scan.close()
