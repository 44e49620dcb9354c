"""Stress run: large perturbations (states far from the nominal, rough Jacobians, large defects); every instance must end either certified
(status 0), interior-point accurate (4) or flagged (1 max-iter, 2 infeasible, 3 numerical) -- never NaN output for a solved instance."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import BatchedFastSLS, make_batch
fx = {"rocket": "sweep_rocket_N20_s0.npz", "quadrotor": "sweep_quadrotor_N20_s0.npz", "pendulum": "sweep_pendulum_N10_s0.npz"}
for model in ("pendulum", "quadrotor", "rocket"):
    for x0_amp, jac, c in ((2.0, 1e-2, 1e-2), (8.0, 3e-2, 5e-2)):
        B = 2048
        batch = make_batch(model, os.path.join(ROOT, "tests", "golden", fx[model]), B, seed=99, x0_amp=x0_amp, jac_amp=jac, c_amp=c)
        m, N = batch["model"], batch["N"]
        f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=B)
        f.set_rti_steps(1)
        f.update_dynamics_list(batch["A"], batch["B"], batch["E"], batch["g"], batch["gN"], batch["c"])
        f.update_linear_cost(batch["q"])
        out = f.solve(batch["x0_arg"])
        st = out["status"]; ok = out["success"]
        finite = np.isfinite(out["primal_vec"][ok]).all() and np.isfinite(out["backoff"][ok]).all()
        kk = out["kkt"][st == 0][:, :3].max() if (st == 0).any() else float("nan")
        print(f"{model:9s} x0_amp {x0_amp} jac {jac} c {c}: status counts {np.bincount(st, minlength=5).tolist()} success {ok.mean():.3f} "
              f"finite {bool(finite)} certificate max {kk:.2e} t_qp {out['t_qp_ms']:.1f} ms")
        f.close()
