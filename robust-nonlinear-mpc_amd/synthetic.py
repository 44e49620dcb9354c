"""Seeded synthetic batches of fast-SLS problem instances (the batch axis = independent MPC instances:
Monte-Carlo initial states / disturbance seeds).

An instance is what SCP_SLS.update_jacobian (solver/SCP_SLS_jit.py:251-366) hands to the fast-SLS path:
A_k, B_k (Jacobians along a nominal), c_k (dynamics defects), g_k = g - G[z_k;v_k], g_N = gf - Gf z_N,
q = 2 H y_nom, x0_arg = x_nom0 - x_meas.  The nominal and its Jacobians come from a fixture file
(tests/golden/sweep_<model>_N*.npz: finite differences of the reference's own RK4 `ddyn`); every instance gets
its own seeded perturbation of the Jacobians, defects and measured state.
"""
import numpy as np

from .models import get_model


def make_batch(model_name, fixture_npz, B, seed=0, x0_amp=0.5, jac_amp=1e-3, c_amp=1e-3):
    m = get_model(model_name)
    g = dict(np.load(fixture_npz))
    N = int(g["N"])
    nx, nu = m.nx, m.nu
    rng = np.random.default_rng(seed)
    A = g["A"][None] + jac_amp * rng.standard_normal((B, N, nx, nx))
    Bm = g["B"][None] + jac_amp * rng.standard_normal((B, N, nx, nu))
    X, U = g["X"], g["U"]
    c = c_amp * rng.standard_normal((B, N, nx))
    Z = np.concatenate([X[:N], U], axis=1)                                  # (N, nz)
    gk = m.g[None, :] - Z @ m.G.T                                           # (N, ni)
    gN = m.gf - m.Gf @ X[N]
    Hd = np.concatenate([np.concatenate([np.diag(m.Q), np.diag(m.R)])] * N + [np.diag(m.Qf)])
    y_nom = np.concatenate([np.concatenate([X[k] - m.x_ref, U[k] - m.u_ref]) for k in range(N)] + [X[N] - m.x_ref])
    q = 2.0 * Hd * y_nom
    scale = 0.5 * (m.x_ub - m.x_lb)
    x0_arg = x0_amp * 0.1 * scale[None] * rng.uniform(-1, 1, (B, nx))
    return dict(model=m, N=N, A=A, B=Bm, E=np.stack([m.E] * (N + 1)), g=np.broadcast_to(gk, (B, N, m.ni)).copy(),
                gN=np.broadcast_to(gN, (B, m.ni_f)).copy(), c=c, q=np.broadcast_to(q, (B, q.size)).copy(), x0_arg=x0_arg)
