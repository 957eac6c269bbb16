#!/bin/bash
# Streamed-ingest rates over the pipeline's knobs (pinned / device pieces, sub-piece size, piece size, producer threads).
# usage: tools/ingest_grid.sh > gpurun_out/ingest_grid.log
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python tools/measure_ingest.py 40000000 2>&1 | grep -E "host memory|table file"; }
run KGWAS_X=default
run KGWAS_INGEST_DEVICE=3
run KGWAS_INGEST_DEVICE=5
run KGWAS_INGEST_PINNED=4
run KGWAS_INGEST_PINNED=6
run KGWAS_INGEST_SUB_MIB=16
run KGWAS_INGEST_SUB_MIB=1
run KGWAS_INGEST_THREADS=4
run KGWAS_INGEST_THREADS=6
run KGWAS_INGEST_THREADS=10
run KGWAS_INGEST_THREADS=12
run KGWAS_INGEST_THREADS=16
run KGWAS_INGEST_PIECE_ROWS=493440
run KGWAS_INGEST_PIECE_ROWS=1973760
