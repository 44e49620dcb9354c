"""Seeded fast-SLS problem instances built from the golden fixtures (test helper, CPU only).

An instance is what SCP_SLS.update_jacobian (solver/SCP_SLS_jit.py:251-366) hands to
fast_SLS.update_dynamics_list / update_linear_cost / solve:
  A_k, B_k   Jacobians of RK4 ddyn along a nominal (finite differences of the reference's own ddyn, stored in
             tests/golden/sweep_*.npz by gen_golden.py)
  c_k        f(z_k, v_k) - z_{k+1}
  g_k        g - G [z_k; v_k],  g_N = gf - Gf z_N
  q          2 H y_nom
  x0_arg     x_nom0 - x_meas
"""
import os

import numpy as np

from conftest import GOLDEN
from robust_nonlinear_mpc_amd import get_model

FIXTURE = {"pendulum": "sweep_pendulum_N10_s0.npz", "quadrotor": "sweep_quadrotor_N20_s0.npz", "rocket": "sweep_rocket_N20_s0.npz"}


class Instance:
    pass


def make_instance(model_name, seed=0, x0_amp=0.2, c_amp=1e-3, N=None):
    """N: horizon; default = the fixture's.  Other horizons reuse the fixture's stages cyclically (data only needs to be plausible)."""
    m = get_model(model_name)
    g = dict(np.load(os.path.join(GOLDEN, FIXTURE[model_name])))
    N0 = int(g["N"])
    if N is not None and N != N0:
        idx = np.arange(N) % N0
        g["A"], g["B"], g["U"] = g["A"][idx], g["B"][idx], g["U"][idx]
        g["X"] = g["X"][np.arange(N + 1) % (N0 + 1)]
    N = N0 if N is None else int(N)
    nx, nu, nz = m.nx, m.nu, m.nz
    rng = np.random.default_rng(1000 + seed)
    inst = Instance()
    inst.m, inst.N = m, N
    inst.A, inst.B = g["A"].copy(), g["B"].copy()
    # perturb the Jacobians a little per seed so that instances differ (batch axis)
    inst.A += 1e-3 * rng.normal(size=inst.A.shape)
    inst.B += 1e-3 * rng.normal(size=inst.B.shape)
    X, U = g["X"].copy(), g["U"].copy()  # nominal (N+1,nx), (N,nu)
    inst.X, inst.U = X, U
    inst.c = c_amp * rng.normal(size=(N, nx))
    G, Gf = m.G, m.Gf
    inst.g_list = [m.g - G @ np.concatenate([X[k], U[k]]) for k in range(N)] + [m.gf - Gf @ X[N]]
    # cost is centred on the neutral state: deviation coordinates
    Hd = np.concatenate([np.concatenate([np.diag(m.Q), np.diag(m.R)])] * N + [np.diag(m.Qf)])
    y_nom = np.concatenate([np.concatenate([X[k] - m.x_ref, U[k] - m.u_ref]) for k in range(N)] + [X[N] - m.x_ref])
    inst.q = 2.0 * Hd * y_nom
    scale = 0.5 * (m.x_ub - m.x_lb)
    inst.x0_arg = x0_amp * 0.1 * scale * rng.uniform(-1, 1, nx)
    inst.E = np.stack([m.E] * (N + 1))
    return inst


def stack(insts, attr):
    return np.stack([np.asarray(getattr(i, attr)) for i in insts])


# ---- runners shared by tests, smoke() and bench.py --------------------------------------------------------
def oracle_dims(inst):
    from oracle import oracle as O
    m = inst.m
    return O.dims_of(m.nx, m.nu, m.nw, inst.N, m.ni, m.ni_f)


def qp1_bounds(inst):
    """l,u (reference row layout incl. x0 rows) of the un-tightened QP, via the oracle's bookkeeping restatement."""
    from oracle import oracle as O
    m = inst.m
    qp = O.OracleQP(oracle_dims(inst), m.G, m.Gf, m.g, m.gf, m.Q, m.R, m.Qf)
    qp.update_dynamics(inst.A, inst.B, inst.g_list)
    qp.offset_constraints(inst.c.T)
    return qp.bounds_with_x0(inst.x0_arg)


def run_oracle_fastsls(inst, rti_steps=1, settings=None, prev_primal=None):
    from oracle import oracle as O
    m = inst.m
    f = O.OracleFastSLS(oracle_dims(inst), m.G, m.Gf, m.g, m.gf, inst.E, m.Q, m.R, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f,
                        settings or O.tight_settings())
    f.set_rti_steps(rti_steps)
    f.update_dynamics_list(inst.A, inst.B, inst.E, inst.g_list, inst.c)
    f.update_linear_cost(inst.q)
    if prev_primal is not None:
        f._prev_primal = prev_primal.copy()
    out = f.solve(inst.x0_arg)
    out["_qp_info"] = f.qp.last_info
    return out


def make_gpu_solver(insts, device=0):
    from robust_nonlinear_mpc_amd import BatchedFastSLS
    m, N = insts[0].m, insts[0].N
    return BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=len(insts), device=device)


def push_instances(f, insts):
    N = insts[0].N
    g = np.stack([np.stack(i.g_list[:N]) for i in insts])
    gN = np.stack([i.g_list[N] for i in insts])
    f.update_dynamics_list(stack(insts, "A"), stack(insts, "B"), insts[0].E, g, gN, stack(insts, "c"))
    f.update_linear_cost(stack(insts, "q"))


def run_gpu_fastsls(insts, rti_steps=1, solver=None):
    f = solver or make_gpu_solver(insts)
    f.set_rti_steps(rti_steps)
    push_instances(f, insts)
    out = f.solve(stack(insts, "x0_arg"))
    if solver is None:
        f.close()
    return out


# ---- closed-loop oracle (restates SCP_SLS.solve / socp_step / reset_warm_start around the oracle fast-SLS) ------------------
def host_ddyn(mid, x, u):
    """RK4 plant step by the INDEPENDENT numpy restatement (oracle/dyn_oracle.py), not by the product's csrc/dynamics.hpp: the
    closed-loop oracle below and the NLP certificate of the GPU tests share no dynamics code with the library under test."""
    from oracle import dyn_oracle as DO
    return DO.ddyn(mid, np.asarray(x, dtype=float), np.asarray(u, dtype=float))


def host_jac(mid, x, u):
    """A, B (complex-step derivatives of the numpy restatement), f = ddyn(x,u)."""
    from oracle import dyn_oracle as DO
    return DO.jac(mid, np.asarray(x, dtype=float), np.asarray(u, dtype=float))


class _Info:
    def __init__(self, ok, its, obj):
        self.status, self.iter, self.polish_status, self.obj_val, self.setup_time_ms, self.solve_time_ms = (1 if ok else -2), its, 0, obj, 0.0, 0.0


def ipm_backend(qp, l, u):
    """Exact solve of the oracle QP object's problem by the dense Mehrotra interior point of tests/ref_ipm.py (nothing in common with the ADMM
    restatement or the GPU solver), returned in the reference's row layout (qp_jit.py:101-123: per stage nx dynamics rows + ni rows of G = [I;-I],
    then the terminal rows, then the x0 pin) with OSQP's sign convention for y.  Box constraints G = [I;-I] only."""
    from ref_ipm import build_equalities, polish, qp_box
    d = qp.d
    nx, nu, N, ni = d.nx, d.nu, d.N, d.ni
    nz, SR = nx + nu, nx + d.ni
    n = nz * N + nx
    hi, lo = np.full(n, 1e20), np.full(n, -1e20)
    for k in range(N):
        hi[k * nz:(k + 1) * nz] = u[k * SR + nx:k * SR + nx + nz]
        lo[k * nz:(k + 1) * nz] = -u[k * SR + nx + nz:k * SR + nx + 2 * nz]
    hi[N * nz:], lo[N * nz:] = u[N * SR:N * SR + nx], -u[N * SR + nx:N * SR + 2 * nx]
    x0val = 0.5 * (l[-nx:] + u[-nx:])
    viol = np.maximum(x0val - hi[:nx], lo[:nx] - x0val).max()          # the reference keeps the stage-0 box rows beside the pin
    hi[:nx], lo[:nx] = 1e20, -1e20
    c = np.stack([-0.5 * (u[k * SR:k * SR + nx] + l[k * SR:k * SR + nx]) for k in range(N)])
    A = qp.A.reshape(N, nx, nx); B = qp.B.reshape(N, nx, nu)
    E, e = build_equalities(A, B, c, x0val)
    Pd = 2.0 * np.concatenate([np.concatenate([np.diag(qp.Q.reshape(nx, nx)), np.diag(qp.R.reshape(nu, nu))])] * N + [np.diag(qp.Qf.reshape(nx, nx))])
    if viol > 1e-9:
        return np.zeros(n), np.zeros(len(l)), _Info(False, 0, 0.0)
    z, nu_, lu, ll, ok, its = qp_box(Pd, qp.q, E, e, lo, hi)
    if ok:      # exact multipliers from the identified active set (the reference's OSQP polishes too); kept only when certified
        zp, nup, lup, llp, polished = polish(Pd, qp.q, E, e, lo, hi, z, lu, ll)
        if polished:
            z, nu_, lu, ll = zp, nup, lup, llp
    y = np.zeros(len(l))
    for k in range(N):
        y[k * SR:k * SR + nx] = nu_[nx * (k + 1):nx * (k + 2)]
        y[k * SR + nx:k * SR + nx + nz] = lu[k * nz:(k + 1) * nz]
        y[k * SR + nx + nz:k * SR + nx + 2 * nz] = ll[k * nz:(k + 1) * nz]
    y[N * SR:N * SR + nx] = lu[N * nz:]; y[N * SR + nx:N * SR + 2 * nx] = ll[N * nz:]
    y[-nx:] = nu_[:nx]
    return z, y, _Info(ok, its, float(0.5 * z @ (Pd * z) + qp.q @ z))


def run_oracle_closed_loop(m, N, x0, steps, rti, sls_steps, W=None, settings=None, scp_eps=1e-10, max_scp_iter=100, X_nom=None, U_nom=None, qp_backend=None):
    """Single-instance CPU closed loop: SCP_SLS.solve (solver/SCP_SLS_jit.py:65-152) with the zero-order roll-out initialiser,
    reset_warm_start (:500-551) and the plant update of the closed-loop scripts."""
    from oracle import oracle as O
    d = O.dims_of(m.nx, m.nu, m.nw, N, m.ni, m.ni_f)
    E = np.stack([m.E] * (N + 1))
    fs = O.OracleFastSLS(d, m.G, m.Gf, m.g, m.gf, E, m.Q, m.R, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, settings or O.tight_settings())
    fs.set_rti_steps(sls_steps)
    if qp_backend == "ipm":
        fs.qp.backend = ipm_backend
    mid = m.model_id
    if X_nom is not None:         # caller's first nominal (the role of IPOPT's in SCP_SLS.solve_nominal_trajectory, SCP_SLS_jit.py:161-188)
        X, U = np.array(X_nom, dtype=float), np.array(U_nom, dtype=float)
    else:
        X = np.zeros((N + 1, m.nx)); U = np.tile(m.u_ref, (N, 1))
        X[0] = x0
        for k in range(N):
            X[k + 1] = host_ddyn(mid, X[k], U[k])
    Hd = np.concatenate([np.concatenate([np.diag(m.Q), np.diag(m.R)])] * N + [np.diag(m.Qf)])
    xm = np.asarray(x0, dtype=float).copy()
    log = dict(state=[], u0=[], nominal_x=[], nominal_u=[], backoff_x=[], success=[], scp_iterations=[], primal_infeasibility=[], oracle_qp_converged=[])
    converge = rti is None or rti <= 0          # SCP_SLS default rti = -1: until |delta|inf < epsilon_convergence (SCP_SLS_jit.py:113-135)
    for i in range(steps):
        if i > 0:
            xN = host_ddyn(mid, X[N], U[N - 1])
            X[:N] = X[1:N + 1].copy(); U[:N - 1] = U[1:N].copy(); X[N] = xN
            fs.reset_solver_to_zeros()
        ok = True
        converged = False
        it_used = 0
        qp_conv = True
        for ii in range(max_scp_iter if converge else rti):
            it_used = ii
            A = np.zeros((N, m.nx, m.nx)); Bm = np.zeros((N, m.nx, m.nu)); c = np.zeros((N, m.nx))
            for k in range(N):
                A[k], Bm[k], f = host_jac(mid, X[k], U[k])
                c[k] = f - X[k + 1]
            g_list = [m.g - m.G @ np.concatenate([X[k], U[k]]) for k in range(N)] + [m.gf - m.Gf @ X[N]]
            y_nom = np.concatenate([np.concatenate([X[k], U[k]]) for k in range(N)] + [X[N]])
            fs.update_dynamics_list(A, Bm, E, g_list, c)
            fs.update_linear_cost(2.0 * Hd * y_nom)
            sol = fs.solve(X[0] - xm)
            ok = bool(sol["success"])
            # fast_SLS.solve ignores the outcome of its LAST forward solve (fast_SLS_jit.py:293, 311): when the oracle's ADMM restatement gives up
            # there (iteration cap at eps 1e-9 on a hard tightened QP; OSQP's own default eps 1e-3 would have returned), `success` stays True and
            # the step carries the previous QP's primal -- not the optimum.  Flagged so that tests do not hold the GPU's exact optimum against it.
            if ok and fs.qp.last_info.status != 1:
                qp_conv = False
            if not ok:
                break
            X = X + sol["primal_x"].T
            U = U + sol["primal_u"].T
            if converge and np.max(np.abs(sol["primal_vec"])) < scp_eps:
                converged = True
                break
        if converge:
            ok = ok and converged
        log["scp_iterations"].append(it_used)
        log["oracle_qp_converged"].append(qp_conv)
        # SCP_SLS.socp_step's primal_infeasibility = max_k,i (f(x_k,u_k) - x_{k+1})_i of the updated nominal (SCP_SLS_jit.py:449-456), signed max
        log["primal_infeasibility"].append(max(float(np.max(host_ddyn(mid, X[k], U[k]) - X[k + 1])) for k in range(N)))
        log["state"].append(X[0].copy()); log["u0"].append(U[0].copy()); log["nominal_x"].append(X.copy()); log["nominal_u"].append(U.copy())
        log["backoff_x"].append(np.array(sol["backoff_x"]) if ok else None); log["success"].append(ok)
        xm = host_ddyn(mid, xm, U[0]) + (m.E @ W[i] if W is not None else 0.0)
    return {k: (np.array(v) if k not in ("backoff_x",) else v) for k, v in log.items()}
