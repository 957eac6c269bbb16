#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; for i in 1 2 3; do env "$@" python tools/measure_ingest.py 40000000 2>&1 | grep -E "table file" | awk "{printf \"%s \", \$4}"; done; echo; }
run KGWAS_X=default
run KGWAS_INGEST_PINNED=4
run KGWAS_INGEST_THREADS=4
run KGWAS_INGEST_THREADS=5
run KGWAS_INGEST_THREADS=6
run KGWAS_INGEST_SCHED=0
run KGWAS_INGEST_PINNED=4 KGWAS_INGEST_THREADS=5
run KGWAS_INGEST_PINNED=4 KGWAS_INGEST_SCHED=0
