"""Session creation time (pinned ring + slots) for the bench shapes: python tools/time_create.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kmersgwas_amd as kg
torch.cuda.set_device(0)
rng = np.random.default_rng(1)
for S, P in ((1024, 101), (2048, 201), (1024, 1)):
    Y = rng.standard_normal((P, S)).astype(np.float32)
    t0 = time.time()
    sc = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, [10001] * P, 5, device=0)
    print(S, P, "create %.3f s" % (time.time() - t0))
    del sc
