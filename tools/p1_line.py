import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = j["p1_scan"]
print("p1: pass %.2f ms | all kernels %.2f ms (%.0f GB/s, %.3f) | filter %.2f ms (%.0f GB/s, %.3f) | replay cpu %.2f ms" % (
    p["ms_per_pass"], p["all_kernels_ms_per_pass"], p["kernels_hbm_GBps"], p["kernels_frac_of_8TBps"],
    p["filter_kernel_ms_per_pass"], p["filter_kernel_hbm_GBps"], p["filter_kernel_frac_of_8TBps"], p["replay_cpu_ms_per_pass"]))
