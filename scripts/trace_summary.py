"""Summarise a rocprofv3 kernel trace of bench.py: per QP-solve (delimited by k_phase(first) launches) wall time and tick count."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = None; seg = []; out = []
for r in rows:
    nm = r["Kernel_Name"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = nm.split("(")[0].split("::")[-1].split("<")[0]
    if short in ("k_ne_fwd", "k_ne_bwd", "k_phase", "k_sweep", "k_eta", "k_tighten", "k_conv", "k_set_bounds", "k_finish"):
        out.append((short, s, e))
# print a compact timeline: consecutive fwd/bwd/phase ticks with durations (us)
i = 0; tick = 0; base = out[0][1]
line = []
for (nm, s, e) in out:
    if nm in ("k_ne_fwd",): tick += 1
    line.append("%s@%.2fms:%.0fus" % (nm[2:], (s - base) / 1e6, (e - s) / 1e3))
print(len(out), "kernels;", tick, "fwd launches")
for j in range(0, len(line), 9):
    print("  ".join(line[j:j + 9]))
