"""One-column scan over a table that fills the HBM (1.2 G rows x 1024 samples): pass and filter times (env knobs vary).
   python tools/p1_large_once.py [rows=1200000000] [passes=4]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S = 1024
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W = 1 + S // 64
Y = make_phenotypes(S, 0, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(rows * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, rows, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y[:1], 10001, mac, device=0)
ts, fs = [], []
for it in range(passes + 1):
    scan.reset(); scan.expect_finish()
    t0 = time.perf_counter()
    scan.feed_device(table.data_ptr(), rows, 0, stream); scan.finish()
    if it:
        ts.append((time.perf_counter() - t0) * 1e3); fs.append(scan.stats()["coarse_kernel_ms"])
b = rows * 8.0 * W
print("%s: pass min %.2f median %.2f ms = %.3f of 8 TB/s; filter %.2f ms = %.3f (%.2f TB/s)" % (
    " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("KGWAS_")) or "default", min(ts), float(np.median(ts)), b / (min(ts) * 1e-3) / 8e12,
    min(fs), b / (min(fs) * 1e-3) / 8e12, b / (min(fs) * 1e-3) / 1e12), flush=True)
