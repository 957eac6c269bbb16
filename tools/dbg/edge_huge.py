import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import kmersgwas_amd as kg
from oracle import binding as ob, oracle_np as onp
from helpers import random_table, phenotypes
S, P = 300, 12
rows = random_table(30_000, S, seed=S * 5 + P, dup_frac=0.2)
col = np.arange(S, dtype=np.uint64)
Y = (phenotypes(S, P - 1, seed=P + 29) * np.float32(float(os.environ.get("SCALE", "1e37")))).astype(np.float32)
mac = onp.min_count(S, 0.05, 5)
topn = 150
exp = ob.associate(rows, S, col, Y, topn, mac, batch_size=7000, threads=3)
sc, kept = ob.scores_dense(rows, S, col, Y, mac)
print("oracle dense: nan", np.isnan(sc[:, kept]).sum(axis=1), "inf", np.isinf(sc[:, kept]).sum(axis=1))
for kern in (kg.KERNEL_COARSE, kg.KERNEL_MFMA):
    scan = kg.AssociationScan(S, col, Y, topn, mac, kernel=kern, chunk_rows=4096)
    scan.feed_host(rows[:11_000], 0)
    scan.feed_host(rows[11_000:], 11_000)
    scan.finish()
    st = scan.stats()
    print("kernel", kern, "pushes", st["heap_pushes"], "oracle", exp["pushes"], "cands", st["candidates"])
    for j in range(P):
        k, s, r = scan.result(j)
        o = exp["per_pheno"][j]
        same = (k == o["kmer"]).all() and s.tobytes() == o["score"].tobytes()
        print("  col", j, "same" if same else "DIFF", "n_inf", np.isinf(s).sum(), "n_nan", np.isnan(s).sum(), "oracle inf", np.isinf(o["score"]).sum(), "nan", np.isnan(o["score"]).sum())
    scan.close()
