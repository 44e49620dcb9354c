import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); q = d["config"]["qp"]
print(sys.argv[2], round(d["ms_per_step"], 2), {k: (round(v["ipm_iters_mean"], 2), round(v["ipm_iters_p99"], 1), v["ipm_iters_max"], round(v["block_solves_mean"], 2), v["block_solves_max"], round(v["solved_frac"], 5)) for k, v in q.items()})
