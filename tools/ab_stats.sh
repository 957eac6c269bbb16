#!/bin/bash
# per-kernel time per pass of the bench step for each variant (env settings), under rocprofv3 --kernel-trace --stats:
#   tools/ab_stats.sh "VAR=a" "KGWAS_LIB=tools/bin/libkgwas_x.so" ...      (AB_BENCH_ARGS as in ab_env.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for v in "$@"; do
  i=$((i+1)); rm -rf gpurun_out/abst$i
  env $v rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/abst$i -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-subrecords $AB_BENCH_ARGS > gpurun_out/abst$i.log 2>&1
  echo "== $v"
  python - gpurun_out/abst$i <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    t=float(r["TotalDurationNs"])/1e6/5
    if t>=0.1 and "synth" not in r["Name"]: print("  %-58s calls/pass %6.1f  %7.3f ms/pass  avg %8.1f us" % (r["Name"][:58], int(r["Calls"])/5, t, float(r["AverageNs"])/1e3))
PY
done
