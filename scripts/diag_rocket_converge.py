"""Diagnostic: rocket SCP converge mode (rti=-1) on the GPU against the CPU restatement for growing caps on the number of SCP iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, get_model
from problems import run_oracle_closed_loop
m = get_model("rocket")
N, B = 20, 3
rng = np.random.default_rng(23)
amp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.04
x0 = np.stack([m.x_ref + amp * (m.x_ub - m.x_lb) * rng.uniform(-1, 1, m.nx) for _ in range(B)])
for sls in (1, None):
    for cap in (20,):
        cl = ClosedLoopMPC(m, N, B, rti=-1, fast_sls_rti_steps=sls)
        cl.f.opts.scp_eps = 1e-8
        cl.f.opts.max_scp_iter = cap
        out = cl.run(x0, 1, None)
        dm = cl.f.get("scp_delta_max", ())
        cl.close()
        for b in range(B):
            t0 = time.time()
            ref = run_oracle_closed_loop(m, N, x0[b], 1, -1, sls, None, scp_eps=1e-8, max_scp_iter=cap)
            ex = np.max(np.abs(out["nominal_trajectory_x"][b].transpose(2, 1, 0) - ref["nominal_x"]))
            eu = np.max(np.abs(out["nominal_trajectory_u"][b].transpose(2, 1, 0) - ref["nominal_u"]))
            print(f"sls {sls} cap {cap} inst {b}: gpu its {out['scp_iterations'][b]} succ {out['success'][b]} dmax {dm[b]:.2e} | ref its {ref['scp_iterations']} succ {ref['success']} | err x {ex:.2e} u {eu:.2e} ({time.time()-t0:.0f}s)", flush=True)
