import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, rows = 1024, 1_200_000_000
W = 1 + S // 64
Y = make_phenotypes(S, 0, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(rows * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, rows, S, 20240601, stream)
torch.cuda.synchronize()
ref = None
for cr in (0, 64 << 20, 128 << 20, 256 << 20, 0):
    scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y[:1], 10001, mac, device=0, chunk_rows=cr)
    ts = []
    for it in range(4):
        scan.reset(); scan.expect_finish()
        t0 = time.perf_counter()
        scan.feed_device(table.data_ptr(), rows, 0, stream); scan.finish()
        ts.append((time.perf_counter() - t0) * 1e3)
    st = scan.stats()
    res = tuple(a.tobytes() for a in scan.result(0))
    if ref is None: ref = res
    print("chunk_rows %4d M: pass %.2f ms (min %.2f) chunks %d kernels %.2f filter %.2f records %d replay_cpu %.2f same %s frac %.3f" % (
        cr >> 20, np.median(ts[1:]), min(ts[1:]), st["chunks"], st["score_kernel_ms"], st["coarse_kernel_ms"], st["candidates"], st["replay_cpu_ms"], res == ref,
        rows * 8.0 * W / (min(ts[1:]) * 1e-3) / 8e12))
    scan.close()
