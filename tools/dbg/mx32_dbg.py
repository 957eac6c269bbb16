import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import kmersgwas_amd as kg
from oracle import binding as ob, oracle_np as onp
from helpers import random_table, phenotypes
os.environ["KGWAS_COARSE_MX"] = "1"; os.environ["KGWAS_MX32"] = "2"
for S, P in [(1024, 63), (1024, 31), (1024, 101), (256, 31), (256, 5), (64, 5), (241, 5), (300, 31), (320, 31)]:
    rows = random_table(30_000, S, seed=S * 11 + P, dup_frac=0.0)
    col = np.arange(S, dtype=np.uint64)
    Y = phenotypes(S, P - 1, seed=P + 13)
    mac = onp.min_count(S, 0.05, 5) if S >= 100 else 1
    topn = 211
    exp = ob.associate(rows, S, col, Y, topn, mac, batch_size=9000, threads=4)
    scan = kg.AssociationScan(S, col, Y, topn, mac, kernel=kg.KERNEL_COARSE, chunk_rows=4096)
    scan.feed_host(rows)
    scan.finish()
    st = scan.stats()
    same = []
    for j in range(P):
        k, s, r = scan.result(j)
        o = exp["per_pheno"][j]
        same.append(len(k) == len(o["kmer"]) and (k == o["kmer"]).all() and s.tobytes() == o["score"].tobytes())
    print("S %d P %d: mx32 %d tiles %s groups %s | tested %d vs %d | pushes %d vs %d | columns ok: %s" % (
        S, P, st["coarse_mx32"], st["coarse_mode_tiles"], st["coarse_mode_lgroups"], st["rows_tested"], exp["tested"], st["heap_pushes"], exp["pushes"],
        "".join("1" if x else "0" for x in same)))
    scan.close()
