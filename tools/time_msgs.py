"""Time the merge's export calls on one session (diagnostics): flat history_above vs history_above_msgs, heaps_export vs
heaps_export_msgs, at the configs[3] shape."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, perms, M = 2048, 200, int(os.environ.get("ROWS", "20000000"))
P = perms + 1
W = 1 + (S + 63) // 64
Y = make_phenotypes(S, perms, 7)
mac = kg.min_count(S, 0.05, 5)
col = np.arange(S, dtype=np.uint64)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), M, M, S, 20240601, st)
torch.cuda.synchronize()
sc = kg.AssociationScan(S, col, Y, 10001, mac, record_history=2, host_threads=int(os.environ.get("THREADS", "8")))
sc.feed_device(table.data_ptr(), M, M, st)
thr = sc.lowest()[0] * 0.999
G = 8
blocks = np.asarray([(P * d) // G for d in range(G + 1)], np.uint64)
col0, ncols = blocks[:-1].copy(), (blocks[1:] - blocks[:-1]).astype(np.uint64)
buf = torch.empty(P * 12000 * 3 + 4096, dtype=torch.int64, pin_memory=True).numpy()
def t(f, n=5):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("history_above      %.2f ms" % t(lambda: sc.history_above(thr)))
print("history_above_msgs %.2f ms" % t(lambda: sc.history_above_msgs(thr, col0, ncols, buf)))
print("history_above_msgs (sizes only) %.2f ms" % t(lambda: sc.history_above_msgs(thr, col0, ncols, None)))
allc = np.arange(P, dtype=np.uint64)
print("heaps_export       %.2f ms" % t(lambda: sc.heaps_export(allc)))
print("heaps_export_msgs  %.2f ms" % t(lambda: sc.heaps_export_msgs(col0, ncols, buf)))
w = sc.heaps_export_msgs(col0, ncols, buf)
o = int(w[0])
from kmersgwas_amd import dist as kdist
m1 = kdist._parse_msg(buf[o:o + int(w[1])], int(ncols[1]))
cols1 = np.arange(int(col0[1]), int(col0[1] + ncols[1]), dtype=np.uint64)
print("heaps_import (%d cols) %.2f ms" % (len(cols1), t(lambda: sc.heaps_import(cols1, *m1))))
print("finish %.2f ms" % t(lambda: (sc.heaps_import(cols1[:1], *kdist._parse_msg(np.concatenate([[1 + 1 + 3 * int(m1[0][0])], m1[0][:1].view(np.int64), m1[1][:int(m1[0][0])].view(np.int64), m1[2][:int(m1[0][0])].view(np.int64), m1[3][:int(m1[0][0])].view(np.int64)]).astype(np.int64), 1)), sc.finish())))
