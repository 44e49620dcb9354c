"""Diagnostic: fast-SLS converge mode (inner loop, MAX_ITER 30, |primal change| <= 1e-3) on one rocket linearisation: GPU vs oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, get_model
from oracle import oracle as O
from problems import host_ddyn, host_jac
m = get_model("rocket")
N, B = 20, 3
amp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
rng = np.random.default_rng(23)
x0 = np.stack([m.x_ref + amp * (m.x_ub - m.x_lb) * rng.uniform(-1, 1, m.nx) for _ in range(B)])
cl = ClosedLoopMPC(m, N, B, rti=-1, fast_sls_rti_steps=None)
cl.f.opts.scp_eps = 1e-8
cl.f.opts.max_scp_iter = 1
out = cl.run(x0, 1, None)
f = cl.f
print("gpu: success", f.get("success", (), np.int32), "iteration_number", f.get("iteration_number", (), np.int32), "status", f.get("status", (), np.int32), "qp_stats", f.get("qp_stats", (2, 8), np.int32)[:, :, [0, 1, 5, 6, 7]].tolist())
gp = f.get("primal_vec", (f.n,))
cl.close()
d = O.dims_of(m.nx, m.nu, m.nw, N, m.ni, m.ni_f)
E = np.stack([m.E] * (N + 1))
Hd = np.concatenate([np.concatenate([np.diag(m.Q), np.diag(m.R)])] * N + [np.diag(m.Qf)])
for b in range(B):
    fs = O.OracleFastSLS(d, m.G, m.Gf, m.g, m.gf, E, m.Q, m.R, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, O.tight_settings())
    fs.set_rti_steps(None)
    X = np.zeros((N + 1, m.nx)); U = np.tile(m.u_ref, (N, 1)); X[0] = x0[b]
    for k in range(N):
        X[k + 1] = host_ddyn(2, X[k], U[k])
    A = np.zeros((N, m.nx, m.nx)); Bm = np.zeros((N, m.nx, m.nu)); c = np.zeros((N, m.nx))
    for k in range(N):
        A[k], Bm[k], fk = host_jac(2, X[k], U[k]); c[k] = fk - X[k + 1]
    g_list = [m.g - m.G @ np.concatenate([X[k], U[k]]) for k in range(N)] + [m.gf - m.Gf @ X[N]]
    y_nom = np.concatenate([np.concatenate([X[k], U[k]]) for k in range(N)] + [X[N]])
    fs.update_dynamics_list(A, Bm, E, g_list, c); fs.update_linear_cost(2.0 * Hd * y_nom)
    # replay the inner loop by hand to log the primal changes
    fs.initialize_backoff()
    prev = None; log = []
    for i in range(30):
        ok = fs.forward_solve(np.zeros(m.nx) + (X[0] - x0[b]))
        if not ok:
            log.append(("QP failed", fs.qp.last_info.status, fs.qp.last_info.iter)); break
        fs.evaluate_dual_eta()
        p = fs.cur["primal_vec"]
        ch = None if prev is None else float(np.max(np.abs(p - prev)))
        prev = p.copy(); log.append(ch)
        if ch is not None and ch <= 1e-3:
            break
        fs.backward_and_tighten()
    print(f"oracle inst {b}: inner changes", [("%.2e" % v if isinstance(v, float) else v) for v in log], "| gpu-vs-oracle last primal", float(np.max(np.abs(gp[b] - prev))) if prev is not None else None)
