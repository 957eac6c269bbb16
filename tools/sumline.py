import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], 'step %.2f med %.2f | tie_heavy %.1f starved %.1f topn %.0f shard %.1f shard2thr %.1f p1 %.3f' % (d['ms_per_step'], d.get('ms_per_step_median', 0), d['tie_heavy']['ms_per_step'], d['starved_host']['ms_per_step'], d['default_topn']['ms_per_step'], d['north_star_shard']['ms_per_step'], d['north_star_shard']['host_threads_2']['ms_per_step'], d['p1_scan']['ms_per_pass']))
