"""Long closed loop in the hard regime: which QP solves are the stragglers of a launch (ticks, interior-point iterations, final status)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, disturbance_stream, get_model
m = get_model("rocket")
B, N, steps = (int(sys.argv[3]) if len(sys.argv) > 3 else 1024), 20, int(sys.argv[2]) if len(sys.argv) > 2 else 30
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
x0 = m.x_ref + scale * (m.extra["x0"] - m.x_ref)
W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(B)], axis=1)
cl = ClosedLoopMPC(m, N, B)
cl.reset(np.tile(x0, (B, 1)), solve_nominal=True, continuation=2 if scale > 0.6 else 1)
for i in range(steps):
    cl.step(W[i], fetch=False)
    qs = cl.f.get("qp_stats", (2, 8), np.int32)
    line = f"step {i:2d}:"
    for slot in (0, 1):
        st, tk, its = qs[:, slot, 6], qs[:, slot, 1], qs[:, slot, 0]
        ran = (st != -1) & (st != 2)
        big = ran & (tk > 30)
        rd = qs[:, slot, 5]
        first = ran & (qs[:, slot, 7] == 0)
        line += (f" | QP{slot+1} ran {ran.mean():.2f} status(0,1,3,4,5) {[(st[ran] == v).sum() for v in (0, 1, 3, 4)]} ticks mean {tk[ran].mean() if ran.any() else 0:.1f} max {tk.max()}"
                 f" rounds(first-attempt ok) p50/p90/p99/max {np.percentile(rd[first], [50, 90, 99, 100]).astype(int).tolist() if first.any() else []} fallbacks {int((ran & (qs[:, slot, 7] != 0)).sum())} >30 ticks: {big.sum()} (status {np.bincount(st[big], minlength=6).tolist() if big.any() else []}, its max {its[big].max() if big.any() else 0})")
    ct = cl.f.get("chain_times", (4,), np.int64) * 1e-5      # ms inside k_rti_chain per instance: QP1, eta + sweep + tightening, QP2, total
    tot = ct[:, 3]
    order = np.argsort(tot)[::-1][:3]
    line += (f" | chain ms p50/p90/p99/max {np.percentile(tot, [50, 90, 99, 100]).round(2).tolist()} slowest: " +
             "; ".join(f"[qp1 {ct[i,0]:.2f} sw {ct[i,1]:.2f} qp2 {ct[i,2]:.2f} | ticks {qs[i,0,1]}/{qs[i,1,1]} st {qs[i,0,6]}/{qs[i,1,6]} its {qs[i,0,0]}/{qs[i,1,0]}]" for i in order))
    t = cl.f.timing_ms()
    print(line + f" | gpu ms: total {t['total']:.2f} qp {t['qp']:.2f} sweep {t['sweep']:.2f} jac {t['jac']:.2f}", flush=True)
cl.close()
