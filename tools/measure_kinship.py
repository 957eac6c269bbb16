"""emma_kinship_kmers accumulation on synthetic tables (BASELINE.json configs[4] shape: 1135 samples): rows resident
in HBM, rows in host memory (kgwas_kinship_feed_host) and rows in a .table file with a hot page cache
(kgwas_kinship_feed_table, what the CLI does)."""
import sys, time, os, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmersgwas_amd as kg


def write_table(base, S, host_words):
    hdr = np.zeros(16, np.uint8)
    hdr[:4] = np.frombuffer(np.uint32(0xDDCCBBAA).tobytes(), np.uint8)
    hdr[4:12] = np.frombuffer(np.uint64(S).tobytes(), np.uint8)
    hdr[12:16] = np.frombuffer(np.uint32(31).tobytes(), np.uint8)
    with open(base + ".table", "wb") as f:
        f.write(hdr.tobytes())
        f.write(host_words.tobytes())
    open(base + ".names", "w").write("".join("s%d\n" % i for i in range(S)))


def run(rows, S, with_host):
    W = 1 + (S + 63) // 64
    t = torch.empty(rows * W, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    kg.synth_rows_device(t.data_ptr(), 0, rows, S, 20240601, stream)
    torch.cuda.synchronize()
    mc = int(np.ceil(S * 0.05))
    feeds = [("HBM-resident", lambda kin: kin.feed_device(t.data_ptr(), rows, stream))]
    if with_host:
        host = t.cpu().numpy().view(np.uint64)
        d = tempfile.mkdtemp(dir="/tmp")
        base = os.path.join(d, "t")
        write_table(base, S, host)
        tbl = kg.KmersTable(base, 31)
        feeds += [("host memory ", lambda kin: kin.feed_host(host)), (".table file ", lambda kin: kin.feed_table(tbl, 0, rows))]
    ref = None
    for name, fn in feeds:
        best = 1e9
        for it in range(3):
            kin = kg.Kinship(S, mc)
            t0 = time.perf_counter()
            fn(kin)
            K, n = kin.matrix()
            dt = time.perf_counter() - t0
            st = kin.stats()
            kin.close()
            best = min(best, dt)
        if ref is None:
            ref = (K.copy(), n)
        assert n == ref[1] and (K == ref[0]).all()
        print("kinship rows=%d S=%d %s: %.1f ms total (kernels %.1f ms) -> %.3f G rows/s end to end, %.1f GB/s of table, "
              "%.1f T pair-updates/s in the kernels, n_used=%d, checksum %d"
              % (rows, S, name, best * 1e3, st["kernel_ms"], rows / best / 1e9, rows * 8 * W / best / 1e9,
                 rows * S * (S - 1) / 2 / (st["kernel_ms"] * 1e-3) / 1e12, n, int(K.sum() % 1000003)))
    if with_host:
        tbl.close()
        os.remove(base + ".table"); os.remove(base + ".names"); os.rmdir(d)


if __name__ == "__main__":
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    run(rows, 1135, True)
    run(rows // 2, 241, False)
