"""Cycle counters of the streaming filter (a library built with -DKGWAS_MXS_PROF=1, tools/mxs_variants.sh prof:"-DKGWAS_MXS_PROF=1"):
KGWAS_LIB=tools/bin/libkgwas_prof.so python tools/mxs_prof.py [samples perms rows]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmersgwas_amd as kg
from kmersgwas_amd import capi
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
perms = int(sys.argv[2]) if len(sys.argv) > 2 else 200
M = int(sys.argv[3]) if len(sys.argv) > 3 else 40_000_000
import torch
W = 1 + (S + 63) // 64
t = torch.empty(M * W, dtype=torch.int64, device="cuda")
kg.synth_rows_device(t.data_ptr(), 0, M, S, 20240601)
rng = np.random.default_rng(7)
y0 = rng.standard_normal(S).astype(np.float32)
Y = np.ascontiguousarray(np.stack([y0] + [rng.permutation(y0) for _ in range(perms)]).astype(np.float32))
col = np.arange(S, dtype=np.uint64)
mac = kg.min_count(S, 0.05, 5)
scan = kg.AssociationScan(S, col, Y, 10001, mac, device=0)
out = (C.c_ulonglong * 8)()
for rep in range(3):
    scan.feed_device(t.data_ptr(), M, 0, torch.cuda.current_stream().cuda_stream)
    scan.finish()
    st = scan.stats()
    capi.lib.kgwas_debug_mxs_prof(out, 1)
    v = list(out)
    if v[3]:
        print("rep %d: filter %.2f ms | per step: %.0f cycles, of which sync wait %.0f (%.1f %%; barrier part %.0f), slab issue %.0f | epilogue %.0f cycles per pass | steps per pass %.1f | kernel cycles per pass %.0f (counter at 100 MHz?)"
              % (rep, st["coarse_kernel_ms"], v[1] / v[3], v[0] / v[3], 100.0 * v[0] / max(v[1], 1), v[6] / v[3], v[7] / v[3], v[2] / max(v[4], 1), v[3] / max(v[4], 1), v[5] / max(v[4], 1)))
    scan.reset()
scan.close()
