import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robust_nonlinear_mpc_amd import BatchedFastSLS, get_model
from oracle import oracle as O
from problems import host_ddyn, host_jac
from ref_ipm import build_equalities, qp_box
m = get_model("rocket")
N, B = 20, 3
rng = np.random.default_rng(23)
x0 = np.stack([m.x_ref + 0.01 * (m.x_ub - m.x_lb) * rng.uniform(-1, 1, m.nx) for _ in range(B)])
b = 1
X = np.zeros((N + 1, m.nx)); U = np.tile(m.u_ref, (N, 1)); X[0] = x0[b]
for k in range(N):
    X[k + 1] = host_ddyn(2, X[k], U[k])
A = np.zeros((N, m.nx, m.nx)); Bm = np.zeros((N, m.nx, m.nu)); c = np.zeros((N, m.nx))
for k in range(N):
    A[k], Bm[k], fk = host_jac(2, X[k], U[k]); c[k] = fk - X[k + 1]
g_list = [m.g - m.G @ np.concatenate([X[k], U[k]]) for k in range(N)] + [m.gf - m.Gf @ X[N]]
Hd = np.concatenate([np.concatenate([np.diag(m.Q), np.diag(m.R)])] * N + [np.diag(m.Qf)])
y_nom = np.concatenate([np.concatenate([X[k], U[k]]) for k in range(N)] + [X[N]])
q = 2.0 * Hd * y_nom
E = np.stack([m.E] * (N + 1))
d = O.dims_of(m.nx, m.nu, m.nw, N, m.ni, m.ni_f)
fs = O.OracleFastSLS(d, m.G, m.Gf, m.g, m.gf, E, m.Q, m.R, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, O.tight_settings())
fs.set_rti_steps(1)
fs.update_dynamics_list(A, Bm, E, g_list, c); fs.update_linear_cost(q)
so = fs.solve(np.zeros(m.nx))
print("oracle: success", so["success"], "|primal|", np.abs(so["primal_vec"]).max(), "qp info", fs.qp.last_info.status, fs.qp.last_info.iter, fs.qp.last_info.polish_status)
for opt in ({}, {"as_first": 0}, {"as_first": 0, "warm_start": 0, "ipm_restart": 0}):
    f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=1)
    f.set_rti_steps(1)
    for k, v in opt.items():
        setattr(f.opts, k, v)
    f.update_dynamics_list(A[None], Bm[None], E, np.stack(g_list[:N])[None], g_list[N][None], c[None]); f.update_linear_cost(q[None])
    sg = f.solve(np.zeros((1, m.nx)))
    ub2 = f.get("ubg", (f.mb,))[0]
    print("gpu", opt, ": success", sg["success"], "|primal|", np.abs(sg["primal_vec"]).max(), "status", sg["status"], "kkt", sg["kkt"][0, :6], "stats", f.get("qp_stats", (2, 8), np.int32)[0].tolist())
    print("   backoff diff vs oracle", np.abs(sg["backoff"][0] - so["backoff"]).max(), "primal diff", np.abs(sg["primal_vec"][0] - so["primal_vec"]).max(), "eta diff", np.abs(sg["eta"][0] - so["eta"]).max())
    f.close()
# arbitrate QP 2 with the dense numpy interior point
SR = m.nx + m.ni; nz = m.nx + m.nu; n = nz * N + m.nx
hi, lo = np.full(n, 1e20), np.full(n, -1e20)
for k in range(N):
    hi[k * nz:(k + 1) * nz] = ub2[k * SR + m.nx:k * SR + m.nx + nz]; lo[k * nz:(k + 1) * nz] = -ub2[k * SR + m.nx + nz:k * SR + m.nx + 2 * nz]
hi[N * nz:], lo[N * nz:] = ub2[N * SR:N * SR + m.nx], -ub2[N * SR + m.nx:N * SR + 2 * m.nx]
hi[:m.nx], lo[:m.nx] = 1e20, -1e20
Ee, ee = build_equalities(A, Bm, c, np.zeros(m.nx))
z, nu, lu, ll, ok, its = qp_box(2.0 * Hd, q, Ee, ee, lo, hi)
print("numpy IPM on the GPU's QP2 bounds: ok", ok, its, "|z|", np.abs(z).max(), "vs gpu", np.abs(z - sg["primal_vec"][0]).max(), "vs oracle", np.abs(z - so["primal_vec"]).max(), "min box width", (hi - lo)[m.nx:].min())
