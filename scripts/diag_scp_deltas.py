"""Diagnostic: |delta|inf per SCP iteration (rocket, rti=-1, one fast-SLS step per iteration), GPU vs oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, get_model
from oracle import oracle as O
from problems import host_ddyn, host_jac
m = get_model("rocket")
N, B = 20, 3
amp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
rng = np.random.default_rng(23)
x0 = np.stack([m.x_ref + amp * (m.x_ub - m.x_lb) * rng.uniform(-1, 1, m.nx) for _ in range(B)])
gd = []
for cap in range(1, 11):
    cl = ClosedLoopMPC(m, N, B, rti=-1, fast_sls_rti_steps=1)
    cl.f.opts.scp_eps = 1e-12
    cl.f.opts.max_scp_iter = cap
    cl.run(x0, 1, None)
    gd.append((cl.f.get("scp_delta_max", ()), cl.f.get("iteration_number", (), np.int32), cl.f.get("qp_stats", (2, 8), np.int32)[:, :, 7].tolist(), cl.f.get("kkt", (8,))[:, :3].max(axis=1)))
    cl.close()
d = O.dims_of(m.nx, m.nu, m.nw, N, m.ni, m.ni_f)
E = np.stack([m.E] * (N + 1))
Hd = np.concatenate([np.concatenate([np.diag(m.Q), np.diag(m.R)])] * N + [np.diag(m.Qf)])
for b in range(B):
    fs = O.OracleFastSLS(d, m.G, m.Gf, m.g, m.gf, E, m.Q, m.R, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, O.tight_settings())
    fs.set_rti_steps(1)
    X = np.zeros((N + 1, m.nx)); U = np.tile(m.u_ref, (N, 1)); X[0] = x0[b]
    for k in range(N):
        X[k + 1] = host_ddyn(2, X[k], U[k])
    od = []
    for ii in range(10):
        A = np.zeros((N, m.nx, m.nx)); Bm = np.zeros((N, m.nx, m.nu)); c = np.zeros((N, m.nx))
        for k in range(N):
            A[k], Bm[k], fk = host_jac(2, X[k], U[k]); c[k] = fk - X[k + 1]
        g_list = [m.g - m.G @ np.concatenate([X[k], U[k]]) for k in range(N)] + [m.gf - m.Gf @ X[N]]
        y_nom = np.concatenate([np.concatenate([X[k], U[k]]) for k in range(N)] + [X[N]])
        fs.update_dynamics_list(A, Bm, E, g_list, c); fs.update_linear_cost(2.0 * Hd * y_nom)
        sol = fs.solve(X[0] - x0[b])
        if not sol["success"]:
            od.append("fail"); break
        X = X + sol["primal_x"].T; U = U + sol["primal_u"].T
        od.append((float(np.max(np.abs(sol["primal_vec"]))), int(sol["iteration_number"])))
    print(f"inst {b}: oracle |delta| (sls its):", [(f"{v[0]:.2e}", v[1]) if not isinstance(v, str) else v for v in od])
    print(f"         gpu    |delta| (sls its):", [(f"{g[0][b]:.2e}", int(g[1][b])) for g in gd], "kkt", [f"{g[3][b]:.1e}" for g in gd])
