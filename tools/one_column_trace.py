"""KGWAS_TRACE of the second pass of a one-column scan (diagnostics)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KGWAS_TRACE"] = "1"
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, M = 1024, 100_000_000
W = 1 + S // 64
Y = make_phenotypes(S, 0, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y[:1], 10001, mac, device=0)
scan.feed_device(table.data_ptr(), M, 0, stream); scan.finish()
scan.reset()
sys.stderr.write("==== second pass\n")
scan.expect_finish()
scan.feed_device(table.data_ptr(), M, 0, stream); scan.finish()
