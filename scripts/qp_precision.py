"""Diagnostic / config-3 sweep: the un-tightened QP (QP#1 of an MPC step, cold start) solved with opts.precision = 0 (fp64) and 1
(fp32 factorisations + fp64 residuals) over a range of interior-point tolerances; reports time, interior-point iterations, block
solves per instance, certified fraction, fp64 re-solves, and the parity of the mixed result against the fp64 one.

    python scripts/qp_precision.py [B] [model]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import BatchedFastSLS, make_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
model = sys.argv[2] if len(sys.argv) > 2 else "quadrotor"
fx = {"rocket": "sweep_rocket_N20_s0.npz", "quadrotor": "sweep_quadrotor_N20_s0.npz", "pendulum": "sweep_pendulum_N10_s0.npz"}[model]
batch = make_batch(model, os.path.join(ROOT, "tests", "golden", fx), B, seed=1)
m, N = batch["model"], batch["N"]
f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=B)
f.update_dynamics_list(batch["A"], batch["B"], batch["E"], batch["g"], batch["gN"], batch["c"])
f.update_linear_cost(batch["q"])
x0 = batch["x0_arg"]
ub = f.get("ubg", (f.mb,)); lb = f.get("lbg", (f.mb,))
l = np.concatenate([lb, -x0 - 1e-10], axis=1); u = np.concatenate([ub, -x0 + 1e-10], axis=1)
f.qp_update_data_vec(batch["q"], l, u)
f.opts.warm_start = 0
ref = None
print(f"{model} N={N} B={B}: cold QP solve")
print("precision  eps     ms      ipm its(mean/max)  solves/inst  factor/inst  certified  fp64-resolved  max rel err vs fp64(1e-9)")
rows = []
for prec, eps in [(0, 1e-9), (0, 1e-6), (1, 1e-3), (1, 1e-4), (1, 1e-5), (1, 1e-6), (1, 1e-7), (1, 1e-9)]:
    f.opts.precision = prec; f.opts.qp_eps = eps
    f.qp_solve(); f.kernel_timing()
    x, y, st, it, t = f.qp_solve()
    f.kernel_timing()
    kk = f.get("kkt", (8,))
    if ref is None:
        ref = x.copy()
    err = np.max(np.abs(x - ref), axis=1) / np.maximum(1.0, np.max(np.abs(ref), axis=1))
    print("%-9s  %.0e  %7.3f  %5.2f / %2d         %6.2f       %6.2f       %.4f     %5d          %.2e" % (
        "fp64" if prec == 0 else "mixed", eps, t * 1e3, it.mean(), it.max(), kk[:, 7].mean(), kk[:, 6].mean(), np.mean(st == 0), f.mx_retries, err.max()))
f.close()
