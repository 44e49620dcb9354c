#!/usr/bin/env python3
"""Generate golden fixtures from the reference's OWN source (runs only in the build container).

What is pinned (partial oracles #1 and #2):
  * the three numba kernels of /root/reference/solver/fast_SLS_jit.py
      _backward_solve_numba (:65-84), _propagate (:87-117), _backoff_from_phi (:120-188)
    executed as plain NumPy (numba.njit stubbed to the identity decorator; fastmath=False, so
    IEEE semantics are the same),
  * Model.ddyn / Pendulum|Quadrotor|Rocket.ode (dyn/model.py:15-34, dyn/*.py) evaluated
    numerically through a NumPy-backed stand-in for the handful of casadi *functions* the ODEs call
    (sin, cos, atan, sqrt, vertcat, hcat, mtimes, cross, inv, diag, vertsplit),
  * the models' constants (dims, g, gf, E) and OCP.riccati_step (solver/ocp.py:103-109).

Nothing of the reference is copied: this script imports it from /root/reference, feeds seeded
inputs, and stores inputs + outputs as .npz under tests/golden/.  /root/reference does not exist on
the GPU box; tests read only the .npz files.

QP solves are NOT pinned here: the arithmetic lives in osqp==1.0.4 (requirements.txt:31), which is
absent from this image ("parity unpinned").
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    sys.path.insert(0, REF)
    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.njit = njit
    nb.prange = range
    sys.modules["numba"] = nb
    pt = types.ModuleType("prettytable")
    pt.PrettyTable = object
    pt.NONE = 0
    pt.HEADER = 1
    sys.modules["prettytable"] = pt
    sys.modules["osqp_generated"] = types.ModuleType("osqp_generated")

    ca = types.ModuleType("casadi")
    ca.sin = np.sin
    ca.cos = np.cos
    ca.atan = np.arctan
    ca.sqrt = np.sqrt

    def vertcat(*a):
        return np.concatenate(
            [np.atleast_1d(np.asarray(x, dtype=float)) if np.ndim(x) < 2 else np.asarray(x, dtype=float) for x in a],
            axis=0,
        )

    ca.vertcat = vertcat
    ca.hcat = lambda l: np.asarray(l, dtype=float).reshape(1, -1)
    ca.mtimes = lambda a, b: np.asarray(a) @ np.asarray(b)
    ca.cross = lambda a, b: np.cross(a, b)
    ca.inv = np.linalg.inv
    ca.diag = lambda v: np.diag(np.asarray(v).ravel())
    ca.vertsplit = lambda X: [X[i] for i in range(len(X))]

    class _Sym:
        def __init__(self, n):
            self._n = n

        def name(self):
            return self._n

    class SX:
        @staticmethod
        def sym(name, *a):
            return _Sym(name)

    ca.SX = SX
    ca.DM = np.asarray
    sys.modules["casadi"] = ca
    import matplotlib

    matplotlib.use("Agg")


def model_setup(name):
    """Model + weights exactly as the reference's closed-loop scripts configure them."""
    from dyn.pendulum import Pendulum
    from dyn.quadrotor import Quadrotor
    from dyn.rocket import Rocket

    if name == "pendulum":  # expe/main_pendulum_robust_closed_loop.py:24-48
        m = Pendulum()
        m.E = 0.003 * np.eye(m.nx)
        x_max = 10 * np.ones(m.nx)
        u_max = 5 * np.ones(m.nu)
        m.replace_constraints(x_max, -x_max, u_max, -u_max, x_max, -x_max)
        Q, R, Qf = np.eye(4), np.eye(1), 10 * np.eye(4)
        regs = (1e3 * np.eye(4), 1e3 * np.eye(1), 1e4 * np.eye(4))
        x_ref = np.zeros(4)
        u_ref = np.zeros(1)
    elif name == "quadrotor":  # expe/main_quadrotor_robust_closed_loop.py:35-69
        m = Quadrotor()
        Q = np.diag([10.0] * 3 + [1.0] * 3 + [1.0] * 4 + [2.0] * 3)
        R = np.eye(4)
        Qf = 10 * Q
        st = np.deg2rad(2.0)
        qv = 0.5 * st
        qw = 0.1 * qv
        m.E = 0.05 * 5 * np.diag([0.10] * 3 + [0.15] * 3 + [qw, qv, qv, qv] + [0.2] * 3)
        regs = (1e4 * np.eye(13), 1e4 * np.eye(4), 1e4 * np.eye(13))
        x_ref = np.asarray(m.neutral_state, dtype=float)
        u_ref = np.asarray(m.neutral_input, dtype=float)
    elif name == "rocket":  # expe/main_rocket_robust_closed_loop.py:32-85
        m = Rocket()
        Q = np.diag([10.0] * 3 + [1.0] * 8 + [5.0, 5.0] + [1.0] * 4)
        R = np.eye(4)
        Qf = 10 * Q
        st = np.deg2rad(2.0)
        qv = 0.5 * st
        qw = 0.1 * qv
        m.E = 0.05 * np.diag([0.20] * 6 + [qv, qv, qv, qw] + [0.2] * 3 + [0.8, 0.2, 0.04, 0.04])
        regs = (1e4 * np.eye(17), 1e4 * np.eye(4), 1e4 * np.eye(17))
        x_ref = np.asarray(m.neutral_state, dtype=float)
        u_ref = np.zeros(4)
    else:
        raise ValueError(name)
    return m, Q, R, Qf, regs, x_ref, u_ref


def fd_jac(m, x, u, h=1e-6):
    nx, nu = x.size, u.size
    A = np.zeros((nx, nx))
    B = np.zeros((nx, nu))
    for i in range(nx):
        e = np.zeros(nx)
        e[i] = h
        A[:, i] = (np.asarray(m.ddyn(x + e, u)) - np.asarray(m.ddyn(x - e, u))) / (2 * h)
    for i in range(nu):
        e = np.zeros(nu)
        e[i] = h
        B[:, i] = (np.asarray(m.ddyn(x, u + e)) - np.asarray(m.ddyn(x, u - e))) / (2 * h)
    return A, B


def random_traj(m, x_ref, u_ref, N, rng, amp):
    nx, nu = m.nx, m.nu
    X = np.zeros((N + 1, nx))
    U = np.zeros((N, nu))
    X[0] = x_ref + amp * rng.uniform(-1, 1, nx)
    for k in range(N):
        U[k] = u_ref + amp * rng.uniform(-1, 1, nu)
        X[k + 1] = np.asarray(m.ddyn(X[k], U[k])).reshape(-1) + 0.01 * amp * rng.uniform(-1, 1, nx)
    return X, U


def sweep_case(F, name, N, seed, general_G=False, nw_cols=None):
    m, Q, R, Qf, regs, x_ref, u_ref = model_setup(name)
    rng = np.random.default_rng(seed)
    nx, nu, nw = m.nx, m.nu, m.nw
    X, U = random_traj(m, x_ref, u_ref, N, rng, 0.1)
    A = np.zeros((N, nx, nx))
    B = np.zeros((N, nx, nu))
    for k in range(N):
        A[k], B[k] = fd_jac(m, X[k], U[k])
    if general_G:
        ni, ni_f = 7, 5
        G = rng.normal(size=(ni, nx + nu))
        Gf = rng.normal(size=(ni_f, nx))
    else:
        G = np.asarray(m.G, dtype=float)
        Gf = np.asarray(m.Gf, dtype=float)
        ni, ni_f = m.ni, m.ni_f
    E = np.stack([np.asarray(m.E, dtype=float)] * (N + 1))
    if nw_cols is not None:      # fewer disturbance channels than states (dyn/LTV.py:17-32 takes any nw): dense nx x nw blocks that differ per stage
        nw = int(nw_cols)
        E = np.stack([np.asarray(m.E, dtype=float)[:, :nw] + 0.01 * rng.normal(size=(nx, nw)) for _ in range(N + 1)])
    eta = np.zeros((N, N, ni))
    for k in range(N):
        for j in range(k + 1):
            mask = rng.uniform(size=ni) < 0.2
            eta[k, j] = mask * 10.0 ** rng.uniform(-2, 5, size=ni)
    eta_f = (rng.uniform(size=(N + 1, ni_f)) < 0.2) * 10.0 ** rng.uniform(-2, 5, size=(N + 1, ni_f))
    Q_reg, R_reg, Q_reg_f = regs
    S, K = F._backward_solve_numba(N, nx, nu, A, B, G, Gf, eta, eta_f, Q_reg, R_reg, Q_reg_f)
    Phi_x, Phi_u = F._propagate(A, B, E, K)
    beta, beta_f, backoff, backoff_f = F._backoff_from_phi(
        Phi_x, Phi_u, np.ascontiguousarray(G[:, :nx]), np.ascontiguousarray(G[:, nx:]), Gf, 1e-10
    )
    d = dict(
        N=N, nx=nx, nu=nu, nw=nw, ni=ni, ni_f=ni_f, A=A, B=B, E=E, G=G, Gf=Gf, eta=eta, eta_f=eta_f,
        Q_reg=Q_reg, R_reg=R_reg, Q_reg_f=Q_reg_f, K=K, beta=beta, beta_f=beta_f, backoff=backoff,
        backoff_f=backoff_f, X=X, U=U,
    )
    if nx <= 4:
        d.update(S=S, Phi_x=Phi_x, Phi_u=Phi_u)
    else:  # keep fixtures small: per-(k,j) Frobenius norms + a weighted checksum
        w = np.cos(np.arange(nx * nx)).reshape(nx, nx)
        d.update(
            S_fro=np.linalg.norm(S, axis=(2, 3)), S_chk=np.einsum("kjab,ab->kj", S, w),
            Phix_fro=np.linalg.norm(Phi_x, axis=(2, 3)), Phix_chk=np.einsum("kjab,ab->kj", Phi_x, w[:, :nw]),
            Phiu_fro=np.linalg.norm(Phi_u, axis=(2, 3)),
        )
    return d


def dyn_case(name, seed, npts=12):
    m, *_, x_ref, u_ref = model_setup(name)
    rng = np.random.default_rng(seed)
    X = x_ref[None] + 0.3 * rng.uniform(-1, 1, (npts, m.nx))
    U = u_ref[None] + 0.3 * rng.uniform(-1, 1, (npts, m.nu))
    ode = np.stack([np.asarray(m.ode(X[i], U[i]), dtype=float).reshape(-1) for i in range(npts)])
    ddyn = np.stack([np.asarray(m.ddyn(X[i], U[i]), dtype=float).reshape(-1) for i in range(npts)])
    A = np.zeros((npts, m.nx, m.nx))
    B = np.zeros((npts, m.nx, m.nu))
    for i in range(npts):
        A[i], B[i] = fd_jac(m, X[i], U[i])
    return dict(
        X=X, U=U, ode=ode, ddyn=ddyn, A_fd=A, B_fd=B, g=np.asarray(m.g, dtype=float), gf=np.asarray(m.gf, dtype=float),
        G=np.asarray(m.G, dtype=float), Gf=np.asarray(m.Gf, dtype=float), E_script=np.asarray(m.E, dtype=float),
        dims=np.array([m.nx, m.nu, m.nw, m.ni, m.ni_f]), x_ref=x_ref, u_ref=u_ref,
    )


def dyn_script_case(name, seed):
    """Points where the closed loops actually run: along roll-outs from the scripts' own initial states
    (expe/main_rocket_robust_closed_loop.py:110-126, main_pendulum...:96, the quadrotor script's random x0 recipe :82-90 with a seeded
    generator), under zero / bound / random bounded inputs, and at saturated actuator states (rocket servo angles +-1, where the
    gimbal linkage's atan/sqrt branch is far from its neutral point)."""
    m, *_, x_ref, u_ref = model_setup(name)
    rng = np.random.default_rng(seed)
    g = np.asarray(m.g, dtype=float)
    nz = m.nx + m.nu
    x_ub, u_ub, x_lb, u_lb = g[:m.nx], g[m.nx:nz], -g[nz:nz + m.nx], -g[nz + m.nx:]
    if name == "rocket":
        x0 = np.array([1.75729, 4.15951, 4.72757, -0.18913, -0.38367, -0.08697, -0.79487, 0.00768, -0.21110, -0.56883,
                       -0.12752, -0.58026, -0.76542, 0.20555, 0.54610, -0.40116, -0.35401])
        starts = [x0, x_ref + 0.5 * (x0 - x_ref), x_ref + 0.3 * (x0 - x_ref)]
        steps = 10
    elif name == "quadrotor":
        D = np.array([2.0] * 3 + [1.0] * 3 + [0.0] + [0.1] * 3 + [0.5] * 3)
        starts = []
        for _ in range(3):
            x = x_ref + D * rng.uniform(-1, 1, m.nx)
            x[6:10] /= np.linalg.norm(x[6:10])
            starts.append(x)
        steps = 10
    else:
        starts = [np.array([0.5, 0.5, 0.0, 0.0]), np.array([-0.5, 1.0, 0.4, -0.3])]
        steps = 15
    X, U = [], []
    for x0 in starts:
        for mode in ("ref", "rand"):
            x = np.array(x0, dtype=float)
            for k in range(steps):
                u = u_ref.copy() if mode == "ref" else u_lb + (u_ub - u_lb) * rng.uniform(0, 1, m.nu)
                X.append(x.copy()); U.append(u)
                x = np.asarray(m.ddyn(x, u), dtype=float).reshape(-1)
                if not np.all(np.isfinite(x)) or np.max(np.abs(x)) > 1e3:
                    break
    # actuator / input corners
    if name == "rocket":
        for s1 in (-1.0, 1.0):
            for s2 in (-1.0, 1.0):
                x = starts[0] + 0.05 * rng.uniform(-1, 1, m.nx)
                x[15], x[16] = s1, s2
                u = np.where(rng.uniform(size=m.nu) < 0.5, u_lb, u_ub)
                X.append(x); U.append(u)
                x = x_ref + 0.1 * rng.uniform(-1, 1, m.nx)
                x[15], x[16] = s1, s2
                x[13] = rng.choice([-50.0, 50.0]) * 0.2
                X.append(x); U.append(np.zeros(m.nu))
    else:
        for _ in range(4):
            X.append(x_ref + 0.5 * rng.uniform(-1, 1, m.nx) * np.minimum(x_ub, 2.0))
            U.append(np.where(rng.uniform(size=m.nu) < 0.5, u_lb, u_ub))
    X, U = np.stack(X), np.stack(U)
    npts = X.shape[0]
    ode = np.stack([np.asarray(m.ode(X[i], U[i]), dtype=float).reshape(-1) for i in range(npts)])
    ddyn = np.stack([np.asarray(m.ddyn(X[i], U[i]), dtype=float).reshape(-1) for i in range(npts)])
    A = np.zeros((npts, m.nx, m.nx))
    B = np.zeros((npts, m.nx, m.nu))
    for i in range(npts):
        A[i], B[i] = fd_jac(m, X[i], U[i])
    return dict(X=X, U=U, ode=ode, ddyn=ddyn, A_fd=A, B_fd=B, dims=np.array([m.nx, m.nu, m.nw, m.ni, m.ni_f]))


def riccati_case(seed):
    from solver.ocp import OCP  # importable as-is (numpy only)

    rng = np.random.default_rng(seed)
    nx, nu = 5, 2
    A = np.eye(nx) + 0.1 * rng.normal(size=(nx, nx))
    B = rng.normal(size=(nx, nu))
    Cx = np.diag(rng.uniform(1, 3, nx))
    Cu = np.diag(rng.uniform(1, 3, nu))
    S = 4 * np.eye(nx)
    Ks, Ss = [], []
    for _ in range(6):
        K, S = OCP.riccati_step(A, B, Cx, Cu, S)
        Ks.append(K)
        Ss.append(S)
    return dict(A=A, B=B, Cx=Cx, Cu=Cu, S0=4 * np.eye(nx), K=np.stack(Ks), S=np.stack(Ss))


def main():
    install_stubs()
    import solver.fast_SLS_jit as F

    cases = [
        ("pendulum", 10, 0, False), ("pendulum", 10, 1, True), ("pendulum", 3, 2, False),
        ("quadrotor", 20, 0, False), ("rocket", 20, 0, False), ("rocket", 5, 1, False),
    ]
    cases = [c + (None,) for c in cases] + [("pendulum", 6, 4, False, 2), ("quadrotor", 8, 3, False, 5)]
    for name, N, seed, gG, nwc in cases:
        d = sweep_case(F, name, N, seed, gG, nwc)
        fn = f"sweep_{name}_N{N}_s{seed}{'_genG' if gG else ''}{'_nw%d' % nwc if nwc else ''}.npz"
        np.savez_compressed(os.path.join(OUT, fn), **d)
        print("wrote", fn, {k: getattr(v, 'shape', v) for k, v in d.items() if k in ('K', 'beta', 'backoff')})
    for name in ("pendulum", "quadrotor", "rocket"):
        np.savez_compressed(os.path.join(OUT, f"dyn_{name}.npz"), **dyn_case(name, 7))
        print("wrote dyn", name)
        d = dyn_script_case(name, 11)
        np.savez_compressed(os.path.join(OUT, f"dyn_{name}_script.npz"), **d)
        print("wrote dyn script", name, d["X"].shape, "max |ddyn|", float(np.abs(d["ddyn"]).max()))
    np.savez_compressed(os.path.join(OUT, "riccati_lq.npz"), **riccati_case(3))
    # numpy legacy RNG stream used by expe/main_rocket_robust_closed_loop.py:30,180 (np.random.seed(0);
    # w = 2*rand(17)-1 per closed-loop step): reproducible with numpy alone, committed for convenience.
    np.random.seed(0)
    W = np.stack([2 * np.random.rand(17) - 1 for _ in range(30)])
    np.savez_compressed(os.path.join(OUT, "rocket_noise_seed0.npz"), W=W)
    print("done")


if __name__ == "__main__":
    main()
