"""Experiment: merit weight rho of the nominal initialiser at the rocket script's x0 (two-stage continuation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, get_model
m = get_model("rocket")
N, B = 15, 2
for rho in (1e1, 1e2, 1e3, 1e4):
    cl = ClosedLoopMPC(m, N, B)
    cl.reset(np.tile(m.extra["x0"], (B, 1)), solve_nominal=True, continuation=2, rho=rho, max_qp=300)
    i = cl.nlp_info[0]
    print(f"rho {rho:.0e}: status {cl.nlp_status.tolist()} accepted {cl.nlp_iterations.tolist()} cost {i[3]:.2f} defect {i[4]:.1e} viol {i[5]:.1e} w {i[0]:.1e} last |d| {i[7]:.1e}")
    cl.close()
