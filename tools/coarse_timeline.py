"""Cycle timeline of the coarse kernel's passes (diagnostic build: tools/coarse_variants.sh tl:"-DKGWAS_COARSE_TIMELINE=<block>",
then KGWAS_LIB=tools/bin/libkgwas_tl.so python tools/coarse_timeline.py). Waves 0 and 4 of the chosen block (the two waves of
SIMD 0) stamp s_memtime at every step start / operands-ready / MFMAs-issued point and around the epilogue parts of the
LAST coarse launch of a pass over the bench workload. The stamps cost a full s_waitcnt before every MFMA phase."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from kmersgwas_amd import capi
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
for it in range(2):
    scan.reset()
    scan.feed_device(table.data_ptr(), M, 0, stream)
torch.cuda.synchronize()
buf = np.zeros(2 * 16 * 64, np.uint64)
fn = capi.lib.kgwas_debug_coarse_timeline
fn.argtypes = [C.c_void_p, C.c_ulonglong]
assert fn(buf.ctypes.data, buf.size) == 0
tl = buf.reshape(2, 16, 64).astype(np.int64)
for w in range(2):
    print("== wave %d" % (w * 4))
    prev_end = None
    for ps in range(16):
        t = tl[w, ps]
        if t[0] == 0 or t[53] == 0:
            continue
        st = [(t[1 + 3 * k], t[2 + 3 * k], t[3 + 3 * k]) for k in range(16)]
        p_phase = [b - a for a, b, c in st]
        m_phase = [c - b for a, b, c in st]
        print("pass %2d: total %6d | gap before %6s | prologue %5d | load+expand (P) sum %5d max %4d [%s] | MFMA issue (M) sum %5d avg %4d [%s] | "
              "row terms %4d tests %4d hits+emit %5d"
              % (ps, t[53] - t[0], "-" if prev_end is None else str(t[0] - prev_end), st[0][0] - t[0], sum(p_phase), max(p_phase),
                 " ".join(str(x) for x in p_phase), sum(m_phase), sum(m_phase) // 16, " ".join(str(x) for x in m_phase), t[51] - t[50], t[52] - t[51], t[53] - t[52]))
        if t[54] or t[55] or t[56]:
            print("         barrier waits: before the MFMA phases %d, after them %d; MFMA issue time %d (sum over the 16 steps)" % (t[54], t[55], t[56]))
        prev_end = t[53]
