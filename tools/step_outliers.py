"""Per-step statistics of configs[1] steps (diagnostics): what differs in the steps that take longer than the others."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes, cgroup_throttle
S, M, perms = 1024, 100_000_000, 100
W = 1 + S // 64
Y = make_phenotypes(S, perms, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
keys = ("chunks", "candidates", "heap_pushes", "gpu_wait_ms", "replay_tail_ms", "dense_ms", "replay_ms", "replay_min_ms", "replay_cpu_ms", "replay_wall_ms", "replay_splits", "score_kernel_ms")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    th0 = cgroup_throttle()
    t0 = time.perf_counter(); scan.reset(); scan.expect_finish()
    scan.feed_device(table.data_ptr(), M, 0, stream); t2 = time.perf_counter()
    scan.finish(); t3 = time.perf_counter()
    st = scan.stats()
    th1 = cgroup_throttle()
    print("step %5.2f feed %5.2f finish %4.2f |" % ((t3 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3), " ".join("%s %s" % (k.replace("_ms", ""), round(st[k], 2) if isinstance(st[k], float) else st[k]) for k in keys), "| thr", th1[0] - th0[0] if th0 and th1 else None)
