"""Would a twisted (two-sided) block factorisation pay?  Per active-set round the first and last stage whose set entry changed (library built with
-DQP_DIAG_SPAN, SLSQP_SO=...), over a rocket closed loop; cost in re-factorised stages of the forward-only recursion (N - first) against a twisted
factorisation whose meeting point p moves to the changed region (span of {p} and the changed stages + 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, disturbance_stream, get_model
m = get_model("rocket")
B, N, steps = 1024, 20, 6
x0 = m.x_ref + 0.3 * (m.extra["x0"] - m.x_ref)
W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(B)], axis=1)
cl = ClosedLoopMPC(m, N, B)
cl.reset(np.tile(x0, (B, 1)), solve_nominal=True)
tot = dict(fwd=0, tw_mid=0, tw_stay=0, tw_near=0, rounds=0, full=0)
hist = np.zeros((N + 1, N + 1), dtype=int)
for i in range(steps):
    cl.step(W[i], fetch=False)
    dg = cl.f.get("qp_diag", (2, 16, 2), np.int32); qs = cl.f.get("qp_stats", (2, 8), np.int32)
    if i == 0:
        continue
    for b in range(B):
        if qs[b, 0, 6] != 0:
            continue
        p = {k: N // 2 for k in ("tw_mid", "tw_stay", "tw_near")}
        tot["full"] += N
        for slot in range(2):
            if qs[b, slot, 6] != 0:
                continue
            for r in range(16):
                kf, kl = dg[b, slot, r]
                if kf < 0:
                    break
                kf = min(kf, N - 1)
                hist[kf, kl] += 1
                tot["rounds"] += 1
                tot["fwd"] += N - kf
                for pol in p:
                    a_, b_ = min(p[pol], kf), max(p[pol], kl)
                    tot[pol] += b_ - a_ + 1
                    if pol == "tw_mid": p[pol] = (kf + kl) // 2
                    elif pol == "tw_near": p[pol] = kf if abs(p[pol] - kf) < abs(p[pol] - kl) else kl      # stay as close as possible to where p was
cl.close()
print("rounds", tot["rounds"], "mandatory full factorisations (stages)", tot["full"])
for k in ("fwd", "tw_mid", "tw_stay", "tw_near"):
    print(f"{k:8s} re-factorised stages per round {tot[k] / max(1, tot['rounds']):.2f}")
span = np.array([[kl - kf + 1 for kl in range(N + 1)] for kf in range(N + 1)])
print("span (last - first + 1) distribution:", {int(sv): int(hist[span == sv].sum()) for sv in range(1, N + 1) if hist[span == sv].sum()})
print("first-stage histogram:", hist.sum(axis=1).tolist())
print("last-stage histogram:", hist.sum(axis=0).tolist())
