"""Streamed kinship accumulation: host memory -> GPU through kgwas_kinship_feed_host (PCIe-inclusive rate).
   python tools/kin_ingest.py [rows=20000000] [S=1135]      (needs a GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1135
W = 1 + (S + 63) // 64
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
host = table.cpu().numpy().view(np.uint64)
mac = kg.min_count(S, 0.05, 5)
ref = None
for name in ("device", "host", "host"):
    kin = kg.Kinship(S, mac)
    t0 = time.perf_counter()
    if name == "device": kin.feed_device(table.data_ptr(), M, stream)
    else: kin.feed_host(host)
    H, n = kin.partials() if hasattr(kin, "partials") else (None, None)
    dt = time.perf_counter() - t0
    print("kinship %-6s feed: %.1f ms = %.1f GB/s of table (%d rows used)" % (name, dt * 1e3, M * W * 8 / dt / 1e9, n if n is not None else -1), flush=True)
    if H is not None:
        if ref is None: ref = H.copy()
        else: assert (H == ref).all()
