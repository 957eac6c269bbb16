"""One-column scan (100 M x 1024, top-10001), wall time per pass over a number of passes: min / median (tools/ab_events.sh)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, M = 1024, 100_000_000
W = 1 + S // 64
Y = make_phenotypes(S, 0, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y[:1], 10001, mac, device=0)
ts = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    scan.reset(); scan.expect_finish()
    t0 = time.perf_counter()
    scan.feed_device(table.data_ptr(), M, 0, stream)
    scan.finish()
    ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[5:])
st = scan.stats()
print("%s: one column, pass min %.3f median %.3f ms; kernels of the last pass %.3f ms (filter %.3f), %d chunks" %
      (os.path.basename(os.environ.get("KGWAS_LIB", "default")), ts[0], ts[len(ts) // 2], st["score_kernel_ms"], st["coarse_kernel_ms"], st["chunks"]))
