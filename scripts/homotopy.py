"""Experiment: reach the script's far-away rocket x0 by continuation in the initial state (x0_s = x_ref + s (x0 - x_ref)), each stage's nominal
NLP started from the previous stage's solution."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, get_model
m = get_model("rocket")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 15
B = 4
cl = ClosedLoopMPC(m, N, B)
X = U = None
for s in [float(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0.25,0.5,0.6,0.7,0.8,0.9,1.0".split(","))]:
    x0 = np.tile(m.x_ref + s * (m.extra["x0"] - m.x_ref), (B, 1))
    cl.reset(x0, X, U, solve_nominal=True, max_qp=200)
    X, U = cl.f.get("nominal_x", (N + 1, m.nx)), cl.f.get("nominal_u", (N, m.nu))
    info = cl.nlp_info[0]
    print(f"s={s}: status {cl.nlp_status.tolist()} accepted {cl.nlp_iterations.tolist()} cost {info[3]:.2f} defect {info[4]:.2e} viol {info[5]:.2e} w {info[0]:.1e} kappa {info[1]:.3f}")
    if (cl.nlp_status != 0).all():
        break
cl.close()
