import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, get_model, disturbance_stream
m = get_model("rocket"); N=20; B=4096; steps=10
x0 = np.tile(m.x_ref + 0.3 * (m.extra["x0"] - m.x_ref), (B,1))
W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(B)], axis=1)
for warm in (1, 0):
    cl = ClosedLoopMPC(m, N, B)
    cl.f.opts.warm_start = warm
    t0=time.perf_counter()
    out = cl.run_on_device(x0, steps, W, solve_nominal=True)
    dt=time.perf_counter()-t0
    print(f"warm_start {warm}: wall {dt:.2f} s, QP {out['t_qp'].sum()/steps:.1f} ms/step, sweep {out['t_riccati'].sum()/steps:.1f} ms/step, solved {out['success'].mean():.3f}")
    cl.close()
