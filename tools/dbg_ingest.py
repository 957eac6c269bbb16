import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from oracle import binding as ob
from bench import make_phenotypes, usable_cpus
S, rows, topn = 1135, int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000, 10001
W = 1 + (S + 63) // 64
Y = make_phenotypes(S, 100, 7)
P = Y.shape[0]
mac = kg.min_count(S, 0.05, 5)
col = np.arange(S, dtype=np.uint64)
table = torch.empty(rows * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, rows, S, 20240601, stream)
torch.cuda.synchronize()
host = table.cpu().numpy().view(np.uint64)
n_chk = min(2_000_000, rows)
exp = ob.associate(host[: n_chk * W].reshape(n_chk, W), S, col, Y, topn, mac, batch_size=10_000_000, threads=min(usable_cpus(), P))
for mx in ("1", "0"):
    os.environ["KGWAS_COARSE_MX"] = mx
    scan = kg.AssociationScan(S, col, Y, topn, mac, device=0)
    scan.feed_device(table.data_ptr(), n_chk, 0, stream)
    scan.finish()
    st = scan.stats()
    bad = []
    for j in range(P):
        k, sc, r = scan.result(j)
        o = exp["per_pheno"][j]
        if not (len(k) == len(o["kmer"]) and (k == o["kmer"]).all() and (r == o["file_row"]).all() and sc.tobytes() == o["score"].tobytes()):
            bad.append(j)
    print("mx", mx, "pushes", st["heap_pushes"], exp["pushes"], "tested", st["rows_tested"], exp["tested"], "bad columns", bad[:10], len(bad), "mx stat", st["coarse_mx"], st["coarse_mode_tiles"], st["coarse_mode_lgroups"])
    scan.close()
