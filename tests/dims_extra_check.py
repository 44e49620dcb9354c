"""Run by test_gpu_parity.py::test_extra_dims_build in a child process with SLSQP_SO pointing at a library built with
-DSLSQP_EXTRA_DIMS="X(6,2)": a fast-SLS RTI step (2 QPs + sweep) of a random (nx, nu) = (6, 2) plant with box constraints against the oracle."""
import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from robust_nonlinear_mpc_amd import BatchedFastSLS

nx, nu, N, B = 6, 2, 8, 5
nz = nx + nu
rng = np.random.default_rng(3)
x_ub, u_ub = np.full(nx, 1.0), np.full(nu, 0.5)
G = np.vstack([np.eye(nz), -np.eye(nz)]); Gf = np.vstack([np.eye(nx), -np.eye(nx)])
g = np.concatenate([x_ub, u_ub, x_ub, u_ub]); gf = np.concatenate([x_ub, x_ub])
m = SimpleNamespace(nx=nx, nu=nu, nw=nx, ni=2 * nz, ni_f=2 * nx, G=G, Gf=Gf, gf=gf, E=0.01 * np.eye(nx))
Q, R, Qf = np.eye(nx), np.eye(nu), 10.0 * np.eye(nx)
Qr, Rr, Qrf = 1e3 * np.eye(nx), 1e3 * np.eye(nu), 1e3 * np.eye(nx)
A = np.eye(nx)[None, None] + 0.15 * rng.normal(size=(B, N, nx, nx))
Bm = 0.4 * rng.normal(size=(B, N, nx, nu))
c = 1e-3 * rng.normal(size=(B, N, nx))
gk = np.broadcast_to(g, (B, N, 2 * nz)).copy(); gN = np.broadcast_to(gf, (B, 2 * nx)).copy()
q = 0.5 * rng.normal(size=(B, nz * N + nx))
x0 = 0.6 * rng.uniform(-1, 1, (B, nx))
E = np.stack([m.E] * (N + 1))
f = BatchedFastSLS(N, Q, R, m, Qf, Qr, Rr, Qrf, batch=B)
f.set_rti_steps(1)
f.update_dynamics_list(A, Bm, E, gk, gN, c)
f.update_linear_cost(q)
out = f.solve(x0)
f.close()
d = O.dims_of(nx, nu, nx, N, 2 * nz, 2 * nx)
worst = 0.0
for b in range(B):
    fo = O.OracleFastSLS(d, G, Gf, g, gf, E, Q, R, Qf, Qr, Rr, Qrf, O.tight_settings())
    fo.set_rti_steps(1)
    fo.update_dynamics_list(A[b], Bm[b], E, list(gk[b]) + [gN[b]], c[b])
    fo.update_linear_cost(q[b])
    r = fo.solve(x0[b])
    assert bool(out["success"][b]) == bool(r["success"]), (b, out["status"][b])
    if r["success"]:
        worst = max(worst, np.max(np.abs(out["primal_vec"][b] - r["primal_vec"])) / max(1.0, np.max(np.abs(r["primal_vec"]))))
        assert np.allclose(out["backoff"][b], r["backoff"], rtol=1e-6, atol=1e-9), b
        assert out["kkt"][b, :3].max() < 1e-8
assert out["success"].any() and worst < 1e-6, worst
print("extra dims ok", worst)
