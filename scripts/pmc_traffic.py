"""Turn the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh into profiles/rNN/pmc_traffic.json (bytes per launch of every
kernel).  FETCH_SIZE / WRITE_SIZE are reported in KiB (rocprofv3); on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section): both the raw and the doubled figure are stored."""
import csv, glob, json, sys, collections
src, dst = sys.argv[1], sys.argv[2]
build = sys.argv[3] if len(sys.argv) > 3 else "?"
cmd = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 2 --warmup 1 --no-cpu --no-secondary"
tail_n = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # the bench's timed region = the last tail_n launches of the dominant kernel
# (0: read it per pass from the bench line that pass printed -- <src>/../pmc_s<slices>_<COUNTER>.log -- the decoupled loop takes a timing-dependent
#  number of rounds = launches per slice)
slices = int(sys.argv[6]) if len(sys.argv) > 6 else 3
extra = sys.argv[7:]                                         # further bench flags the passes ran with (none for the default command)
DOM = "k_rti_chain"
per_disp = collections.defaultdict(lambda: collections.defaultdict(dict))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0].replace("void ", "").strip()
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
            d = per_disp[k][r["Counter_Name"]]
            d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
out = {}
for k, v in acc.items():
    nf, nw = max(1, len(calls[k]["FETCH_SIZE"])), max(1, len(calls[k]["WRITE_SIZE"]))
    fetch, write = v.get("FETCH_SIZE", 0.0) * 1024 / nf, v.get("WRITE_SIZE", 0.0) * 1024 / nw
    out[k] = {"launches": nf, "fetch_bytes_per_launch_raw": fetch, "fetch_bytes_per_launch_x2": 2 * fetch, "write_bytes_per_launch": write}
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), command: " + cmd, "build": build, "kernels": out}
# the command dictionary bench.py's bench_command() must equal for the figure to be attached to a bench line: the default command, with --steps / --warmup
# taken from the pass's command line
import re
def _flag(name, default):
    m = re.search(r"--" + name + r"\s+(\S+)", cmd)
    return type(default)(m.group(1)) if m else default
if not [e for e in extra if e not in ("--steps", "--warmup") and not e.isdigit()]:
    res["command"] = {"model": "rocket", "batch": 4096, "steps": _flag("steps", 30), "warmup": _flag("warmup", 1), "slices": slices, "x0_scale": 1.0, "precision": 0,
                      "workload": "closed_loop", "decoupled": 1, "round_budget_ms": 8.0, "round_cut_frac": 0.0}
if DOM not in out:
    DOM = "k_qp_solve"
for kn in ("k_rti_chain", "k_qp_solve", "k_sweep_prop", "k_sweep_ric1", "k_lin_tan"):
    if kn in out:
        res[kn + "_bytes_per_launch"] = out[kn]["fetch_bytes_per_launch_x2"] + out[kn]["write_bytes_per_launch"]
if DOM in per_disp:
    # same launches as bench.py's roofline averages: the last tail_n dispatches of the dominant kernel (the earlier ones belong to the untimed set-up / warm-up)
    tot = 0.0
    for cn, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        dd = per_disp[DOM].get(cn, {})
        n_tail = tail_n
        if n_tail <= 0:
            import os
            try:
                line = [l for l in open(os.path.join(os.path.dirname(src.rstrip("/")), f"pmc_s{slices}_{cn}.log")) if l.startswith("{")][-1]
                cfg = json.loads(line)["config"]
                n_tail = sum(cfg["rounds_per_slice"]) if cfg.get("rounds_per_slice") else 30 * slices
            except Exception:
                n_tail = 30 * slices
            res.setdefault("timed_region_launches_per_pass", {})[cn] = n_tail
        last = sorted(dd)[-n_tail:]
        tot += mult * 1024.0 * sum(dd[i] for i in last) / max(1, len(last))
    res[DOM + "_bytes_per_launch_timed_region"] = tot
    res["timed_region_launches"] = tail_n
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res)[:1500])
