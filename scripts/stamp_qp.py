"""Cycle breakdown of k_qp_solve per instance (build with -DQP_STAMP; SLSQP_SO=...): forward sweeps that factorise / that only substitute,
backward sweeps, phase logic -- averaged over the instances of a closed-loop step (kkt slots 2..7 carry the stamps in that build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, disturbance_stream, get_model
m = get_model("rocket")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
N, steps = 20, int(sys.argv[3]) if len(sys.argv) > 3 else 4
x0 = m.x_ref + scale * (m.extra["x0"] - m.x_ref)
W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(B)], axis=1)
cl = ClosedLoopMPC(m, N, B)
cl.reset(np.tile(x0, (B, 1)), solve_nominal=True, continuation=2 if scale > 0.6 else 1)
for i in range(steps):
    cl.step(W[i], fetch=False)
    kk = cl.f.get("kkt", (8,)); qs = cl.f.get("qp_stats", (2, 8), np.int32)
    if not (kk[:, 7] > 0).any():
        continue
    kk, qs = kk[kk[:, 7] > 0], qs[kk[:, 7] > 0]
    tot = kk[:, 7].mean()
    print(f"step {i} QP2: ticks {qs[:,1,1].mean():.2f} factor stages {kk[:,3].mean():.1f} | cycles/instance total {tot:.0f}: fwd-factor {kk[:,2].mean()/tot:.2f} fwd-solve {kk[:,4].mean()/tot:.2f} "
          f"bwd {kk[:,5].mean()/tot:.2f} phase {kk[:,6].mean()/tot:.2f} | per factorised stage {kk[:,2].sum()/max(1,kk[:,3].sum()):.0f} cyc, per bwd sweep {kk[:,5].sum()/qs[:,1,1].sum():.0f}, per phase {kk[:,6].sum()/qs[:,1,1].sum():.0f}", flush=True)
cl.close()
