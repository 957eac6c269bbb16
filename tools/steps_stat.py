"""Headline step (configs[1]) over many passes with the library named by KGWAS_LIB: median, lower quartile, minimum (A/B runs on noisy boxes)."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ts, ks = [], []
for i in range(n + 5):
    t0 = time.perf_counter(); scan.reset(); scan.expect_finish()
    scan.feed_device(table.data_ptr(), M, 0, stream)
    scan.finish()
    ts.append((time.perf_counter() - t0) * 1e3)
    ks.append(scan.stats()["score_kernel_ms"])
ts = sorted(ts[5:]); ks = sorted(ks[5:])
print("%s %s: step median %.2f q25 %.2f min %.2f mean %.2f ms | all kernels median %.2f" % (os.path.basename(os.environ.get("KGWAS_LIB", "default")), sys.argv[2] if len(sys.argv) > 2 else "",
      ts[len(ts) // 2], ts[len(ts) // 4], ts[0], sum(ts) / len(ts), ks[len(ks) // 2]))
