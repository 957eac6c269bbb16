import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, rows = 1024, 100_000_000
W = 1 + S // 64
Y = make_phenotypes(S, 0, 7)
table = torch.empty(rows * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, rows, S, 20240601, stream)
torch.cuda.synchronize()
for cp in (False, True):
    scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, kg.min_count(S, 0.05, 5), device=0, count_patterns=cp)
    ts = []
    for it in range(3):
        scan.reset(); t0 = time.perf_counter(); scan.feed_device(table.data_ptr(), rows, 0, stream); torch.cuda.synchronize(); t1 = time.perf_counter(); scan.finish(); ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
    print(os.environ.get("KGWAS_LIB", "default")[-24:], "count_patterns", cp, "feed %.2f ms finish %.2f ms" % min(ts))
    scan.close()
