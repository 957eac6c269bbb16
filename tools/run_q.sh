python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --samples 2048 --perms 200 --rows 100000000 --steps 5 --warmup 2 --no-cpu-baseline --no-subrecords > gpurun_out/l.json 2>/dev/null
python - <<PY
import json,sys
j=json.loads(open("gpurun_out/l.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("c3 default: step %.1f kernels %.2f coarse %.2f" % (j["ms_per_step"], r["all_scoring_kernels_ms_per_step"], r["kernel_ms_per_step"]), r["kernel"], r["frac"], r["frac_of_int8_fp8_peak_5POPs"], [(c["slices"], c["tiles_per_lds_group"], c["lds_groups"], round(c["ms_per_step"],1)) for c in r["coarse_sets"]])
PY
