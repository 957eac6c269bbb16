"""The `ingest` sub-record of bench.py alone (streamed paths: host memory and .table file against the HBM-resident scan)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kmersgwas_amd as kg
from bench import ingest_record, usable_cpus
torch.cuda.set_device(0)
r = ingest_record(kg, torch, torch.cuda.current_stream().cuda_stream, 0, usable_cpus(), rows=int(os.environ.get("INGEST_ROWS", "40000000")))
print("ingest: hbm %.1f ms | host %.1f ms %.1f GB/s | file %.1f ms %.1f GB/s | parity %s" % (
    r["hbm_resident"]["ms"], r["host_memory"]["ms"], r["host_memory"]["GBps"], r["table_file_page_cache"]["ms"],
    r["table_file_page_cache"]["GBps"], r["parity_check"]))
