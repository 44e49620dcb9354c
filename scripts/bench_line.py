"""Print the few numbers of a bench.py JSON line that matter when tuning (stdin: the line)."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step %.2f  qp avg %.2f  fwd launches %d avg %.3f  sweep %.2f  polished %.4f cold-fallback %.4f its max %d" % (
    d["ms_per_step"], r["qp_solve"]["avg_ms"], r["launches"], r["avg_launch_ms"], r["sweep_avg_launch_ms"], d["config"]["polished_frac"],
    d["config"]["qp2_cold_fallback_frac"], d["config"]["ipm_iters_max_last_qp"]), "slices", d["config"].get("slices_per_gpu"), "frac %.4f" % r["frac"])
