"""Sum the SQ counters of a `rocprofv3 --pmc ...` pass per kernel (profiles/r01/pmc_sq.json).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md, constants table)."""
import collections, csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py --no-cpu --no-secondary --slices 1"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0].replace("void ", "").strip()
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
out = {}
for k, v in acc.items():
    d = dict(launches=len(calls[k]), **{c: x for c, x in v.items()})
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        d["valu_active_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
        d["lds_active_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_LDS", 0.0) / wc
        d["wait_any_frac_of_wave_cycles"] = v.get("SQ_WAIT_ANY", 0.0) / wc
        d["wait_inst_any_frac_of_wave_cycles"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
    if v.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
        d["lds_bank_conflict_frac_of_lds_active"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
    bc = v.get("SQ_BUSY_CYCLES", 0.0)
    if bc > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        d["mfma_busy_cycles_per_sq_busy_cycle"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / bc
    out[k] = d
json.dump({"source": "rocprofv3 --pmc (one SQ pass), command: " + cmd, "kernels": out}, open(dst, "w"), indent=1)
for k in ("k_cl_loop", "k_rti_chain", "k_qp_solve", "k_sweep_prop", "k_sweep_ric1", "k_lin_tan", "k_lin_val"):
    if k in out:
        print(k, {a: (round(b, 4) if isinstance(b, float) and b < 10 else b) for a, b in out[k].items()})
