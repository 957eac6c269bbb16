cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in nt; do
  export KGWAS_LIB=$PWD/tools/bin/libkgwas_$v.so
  rm -rf gpurun_out/pq_$v; KGWAS_MXS=2 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d gpurun_out/pq_$v -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 2 --warmup 1 --no-cpu-baseline --no-subrecords > gpurun_out/pq_$v.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/pq_$v/*/*_counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "mxs_kernel" in r["Kernel_Name"] and int(r["Grid_Size"])==1048576 and r["Counter_Name"]=="FETCH_SIZE"]
print("$v", len(v), "FETCH KiB mean %.0f -> x algorithmic %.3f" % (sum(v)/len(v), 2*sum(v)/len(v)*1024/(8388608*264)))
PY
done
unset KGWAS_LIB
KGWAS_MXS=2 bash tools/mxs_ablate.sh "--samples 2048 --perms 200 --rows 100000000" nont nt 2>&1 | tail -4
