"""Does the interpreter's garbage collector account for the headline loop's late steps? 300 steps with gc callbacks timed, then 300 with
the collector disabled (diagnostics for bench.py's timed region)."""
import sys, os, time, gc
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
gc_ms, t_gc = [], [0.0]
def cb(phase, info):
    if phase == "start": t_gc[0] = time.perf_counter()
    else: gc_ms.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3))
gc.callbacks.append(cb)
def run(n, label):
    ts, keep = [], []
    for i in range(n):
        g0 = len(gc_ms)
        t0 = time.perf_counter(); scan.reset(); scan.expect_finish()
        scan.feed_device(table.data_ptr(), M, 0, stream); scan.finish()
        keep.append(scan.stats())
        ts.append(((time.perf_counter() - t0) * 1e3, [x for x in gc_ms[g0:]]))
    v = sorted(t for t, _ in ts)
    print("%s: mean %.2f median %.2f max %.2f; steps above median + 2 ms: %s" % (label, sum(v) / len(v), v[len(v) // 2], v[-1],
          [(round(t, 1), [(g, round(m, 1)) for g, m in gcs]) for t, gcs in ts if t > v[len(v) // 2] + 2.0]))
run(20, "warm-up")
run(300, "collector on")
gc.collect(); gc.disable()
run(300, "collector off")
