"""Wall-time breakdown of one bench step (reset / feed_device / finish / result) on the BASELINE configs[1] workload."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
for it in range(int(os.environ.get("STEPS", "6"))):
    t = [time.perf_counter()]
    scan.reset(); t.append(time.perf_counter())
    if os.environ.get("EXPECT"): scan.expect_finish()
    scan.feed_device(table.data_ptr(), M, 0, stream); t.append(time.perf_counter())
    scan.finish(); t.append(time.perf_counter())
    st = scan.stats(); t.append(time.perf_counter())
    d = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
    print("step %d: reset %.2f feed %.2f finish %.2f stats %.2f | replay %.1f dense %.1f gpu_wait %.2f kernels %.1f tail %.2f | selected %d replayed_at_finish %d popped_ahead %d pushes %d chunks %d" %
          (it, d[0], d[1], d[2], d[3], st["replay_ms"], st["dense_ms"], st["gpu_wait_ms"], st["score_kernel_ms"], st["replay_tail_ms"],
           st["columns_selected"], st["columns_replayed_at_finish"], st["columns_popped_ahead"], st["heap_pushes"], st["chunks"]))
