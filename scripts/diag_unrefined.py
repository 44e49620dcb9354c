"""How accurate is the un-refined active-set solve?  Distribution of the equality residual the refinement's forward sweep measures (kkt[5]) and of the
accepted solution's certificate quantities, relative to max(1,|q|inf), over a rocket closed loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, disturbance_stream, get_model
m = get_model("rocket")
B, N, steps = 1024, 20, 4
x0 = m.x_ref + 0.3 * (m.extra["x0"] - m.x_ref)
W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(B)], axis=1)
cl = ClosedLoopMPC(m, N, B)
cl.reset(np.tile(x0, (B, 1)), solve_nominal=True)
n = (m.nx + m.nu) * N + m.nx
for i in range(steps):
    cl.step(W[i], fetch=False)
    kk = cl.f.get("kkt", (8,)); q = cl.f.get("q", (n,)); qs = cl.f.get("qp_stats", (2, 8), np.int32)
    sc = np.maximum(1.0, np.abs(q).max(axis=1))
    ran = qs[:, 1, 6] == 0
    pr = lambda v: " ".join(f"{x:.1e}" for x in np.percentile(v[ran] / sc[ran], [50, 90, 99, 100]))
    print(f"step {i}: ran {ran.mean():.2f} | eqres(unrefined) p50/90/99/max {pr(kk[:,5])} | accepted stationarity {pr(kk[:,0])} box {pr(np.maximum(kk[:,1],0))} sign {pr(np.maximum(kk[:,2],0))} | ticks {qs[ran,1,1].mean():.2f}", flush=True)
cl.close()
