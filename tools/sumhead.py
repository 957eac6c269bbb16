import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r = d['roofline']
print(sys.argv[1], 'step mean %.2f median %.2f | mx %.2f all kernels %.2f | chunks %s records %s' % (d['ms_per_step'], d.get('ms_per_step_median', 0), r['kernel_ms_per_step'], r['all_scoring_kernels_ms_per_step'], d['host'].get('chunks_per_step'), d['host'].get('candidates_per_step')))
