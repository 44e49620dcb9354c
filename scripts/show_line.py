"""One-line summary of a bench line file: show_line.py <file> [tag]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
q = d["config"].get("qp", {})
pl = (d["config"].get("persistent_launch") or [{}])[0]
print(sys.argv[2] if len(sys.argv) > 2 else "", round(d["ms_per_step"], 2), "ms/step", round(d["value"]), "QP/s  waves busy", round(pl.get("wave_busy_frac", 0.0), 3),
      {k: (round(v["cold_fallback_frac"], 4), round(v["block_solves_mean"], 2), v["block_solves_max"], round(v["solved_frac"], 5)) for k, v in q.items()})
