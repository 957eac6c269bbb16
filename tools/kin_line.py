"""Kinship sub-record of bench.py alone (8M rows x 1135 accessions)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kmersgwas_amd as kg
from bench import kinship_record
torch.cuda.set_device(0)
r = kinship_record(kg, torch, torch.cuda.current_stream().cuda_stream, 0, cpu_rows=int(os.environ.get("KIN_CPU_ROWS", "4000")))
print("kinship: kernels %.2f ms  wall %.2f ms  %.0f TOP/s  frac %.3f  parity %s  cpu %.0f rows/s" % (
    r["kernels_ms"], r["wall_ms"], r["algorithmic_TOPs"], r["frac"], r["parity_check"], r["cpu_baseline"]["value"]))
