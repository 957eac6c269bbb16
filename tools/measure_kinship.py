"""emma_kinship_kmers accumulation on synthetic tables (BASELINE.json configs[4] shape: 1135 samples)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmersgwas_amd as kg

def run(rows, S):
    W = 1 + (S + 63) // 64
    t = torch.empty(rows * W, dtype=torch.int64, device="cuda")
    kg.synth_rows_device(t.data_ptr(), 0, rows, S, 20240601, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    mc = int(np.ceil(S * 0.05))
    for it in range(2):
        kin = kg.Kinship(S, mc)
        t0 = time.perf_counter()
        kin.feed_device(t.data_ptr(), rows, torch.cuda.current_stream().cuda_stream)
        K, n = kin.matrix()
        dt = time.perf_counter() - t0
        st = kin.stats()
        kin.close()
    print("kinship rows=%d S=%d: %.1f ms total (kernels %.1f ms) -> %.3f G rows/s, %.1f T pair-updates/s, n_used=%d, checksum %d"
          % (rows, S, dt * 1e3, st["kernel_ms"], rows / (st["kernel_ms"] * 1e-3) / 1e9, rows * S * (S - 1) / 2 / (st["kernel_ms"] * 1e-3) / 1e12, n, int(K.sum() % 1000003)))

if __name__ == "__main__":
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    run(rows, 1135)
    run(rows // 2, 241)
