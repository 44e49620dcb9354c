"""Turn the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh into profiles/rNN/pmc_traffic.json (bytes per launch of every
kernel).  FETCH_SIZE / WRITE_SIZE are reported in KiB (rocprofv3); on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section): both the raw and the doubled figure are stored."""
import csv, glob, json, sys, collections
src, dst = sys.argv[1], sys.argv[2]
build = sys.argv[3] if len(sys.argv) > 3 else "?"
cmd = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 2 --warmup 1 --no-cpu --no-secondary"
variant = sys.argv[5] if len(sys.argv) > 5 else "p"        # profile_round.sh's variant tag: the pass's own bench line is <src>/../pmc_<variant>_<COUNTER>.log
# The bench's timed region = the last launches of the dominant kernel, as many as that bench line reports (config.rounds_per_slice: the persistent launch
# is ONE; the round-based loop takes a timing-dependent number of rounds = launches per slice).  The command dictionary (what bench.py's bench_command()
# must equal for the figure to be attached to a bench line) is taken from the same line.
import os
def pass_line(cn):
    try:
        return json.loads([l for l in open(os.path.join(os.path.dirname(src.rstrip("/")), f"pmc_{variant}_{cn}.log")) if l.startswith("{")][-1])
    except Exception:
        return None
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
line0 = pass_line("FETCH_SIZE")
DOM = line0["roofline"]["kernel"] if line0 else next((k for k in ("k_cl_loop", "k_rti_chain", "k_qp_solve") if k in out), "k_qp_solve")
if line0 and line0.get("roofline", {}).get("command"):
    res["command"] = line0["roofline"]["command"]
for kn in ("k_cl_loop", "k_rti_chain", "k_qp_solve", "k_sweep_prop", "k_sweep_ric1", "k_lin_tan"):
    if kn in out:
        res[kn + "_bytes_per_launch"] = out[kn]["fetch_bytes_per_launch_x2"] + out[kn]["write_bytes_per_launch"]
if DOM in per_disp:
    tot = 0.0
    for cn, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        dd = per_disp[DOM].get(cn, {})
        line = pass_line(cn)
        cfg = line["config"] if line else {}
        n_tail = sum(cfg["rounds_per_slice"]) if cfg.get("rounds_per_slice") else 1
        res.setdefault("timed_region_launches_per_pass", {})[cn] = n_tail
        last = sorted(dd)[-n_tail:]
        tot += mult * 1024.0 * sum(dd[i] for i in last) / max(1, len(last))
    res[DOM + "_bytes_per_launch_timed_region"] = tot
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res)[:1500])
