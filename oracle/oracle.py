"""CPU oracle (TEST INFRASTRUCTURE -- never imported by the product package).

ctypes front-end of oracle/libsls_oracle.so plus a single-instance Python restatement of the
bookkeeping the reference does around its kernels (citations relative to /root/reference):

  OracleQP        <- class QP            solver/qp_jit.py:22   (bounds/cost bookkeeping :194-273, :362-402,
                                                                 :487-513, :578-628)
  OracleFastSLS   <- class fast_SLS      solver/fast_SLS_jit.py:195 (solve :278-312, _step :314-327,
                                                                 evaluate_dual_eta :475-487,
                                                                 update_tightening :517-571,
                                                                 check_convergence_socp :581-600,
                                                                 post_processing_solution :602-646)

The QP arithmetic itself (osqp==1.0.4) is restated in sls_oracle.c: "parity unpinned" (see its header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EPSILON = 1e-10  # qp_jit.py:19
BIG = 1e20  # qp_jit.py:382


class Dims(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("nx", "nu", "nw", "N", "ni", "ni_f")]


class OsqpSettings(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf", "eps_dual_inf", "delta")] + [
        (k, C.c_int) for k in ("max_iter", "check_termination", "scaling", "adaptive_rho", "adaptive_rho_interval", "polish", "polish_refine_iter")
    ] + [("adaptive_rho_tolerance", C.c_double)]


class OsqpInfo(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("status", "iter", "rho_updates", "polish_status")] + [
        (k, C.c_double) for k in ("obj_val", "pri_res", "dua_res", "rho_final", "setup_time_ms", "solve_time_ms")
    ]


def build(force=False):
    so = os.path.join(_HERE, "libsls_oracle.so")
    src = os.path.join(_HERE, "sls_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsls_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dims_of(nx, nu, nw, N, ni, ni_f):
    return Dims(nx, nu, nw, N, ni, ni_f)


# ----------------------------------------------------------------------------------------------------
# SLS sweep (fast_SLS_jit.py:65-188)
# ----------------------------------------------------------------------------------------------------
def backward(d, A, B, G, Gf, eta, eta_f, Q_reg, R_reg, Q_reg_f):
    A, B, G, Gf, eta, eta_f, Q_reg, R_reg, Q_reg_f = map(_c, (A, B, G, Gf, eta, eta_f, Q_reg, R_reg, Q_reg_f))
    S = np.zeros((d.N + 1, d.N + 1, d.nx, d.nx))
    K = np.zeros((d.N, d.N + 1, d.nu, d.nx))
    lib().so_backward(C.byref(d), _p(A), _p(B), _p(G), _p(Gf), _p(eta), _p(eta_f), _p(Q_reg), _p(R_reg), _p(Q_reg_f), _p(S), _p(K))
    return S, K


def propagate(d, A, B, E, K):
    A, B, E, K = map(_c, (A, B, E, K))
    Px = np.zeros((d.N + 1, d.N + 1, d.nx, d.nw))
    Pu = np.zeros((d.N, d.N + 1, d.nu, d.nw))
    lib().so_propagate(C.byref(d), _p(A), _p(B), _p(E), _p(K), _p(Px), _p(Pu))
    return Px, Pu


def backoff(d, Px, Pu, G, Gf, eps=1e-10):
    Px, Pu, G, Gf = map(_c, (Px, Pu, G, Gf))
    beta = np.zeros((d.N, d.N, d.ni))
    beta_f = np.zeros((d.N + 1, d.ni_f))
    bo = np.zeros((d.N, d.ni))
    bof = np.zeros(d.ni_f)
    lib().so_backoff(C.byref(d), _p(Px), _p(Pu), _p(G), _p(Gf), C.c_double(eps), _p(beta), _p(beta_f), _p(bo), _p(bof))
    return beta, beta_f, bo, bof


# ----------------------------------------------------------------------------------------------------
# QP (OSQP-class restatement)
# ----------------------------------------------------------------------------------------------------
def default_settings(**kw):
    s = OsqpSettings()
    lib().so_osqp_default_settings(C.byref(s))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def tight_settings(**kw):
    """eps 1e-9 / 50000 its: what qp_jit.py:287-306 asks for at construction (discarded by the reference
    before the first solve).  Used by tests to drive the restatement to the exact optimum."""
    return default_settings(eps_abs=1e-9, eps_rel=1e-9, max_iter=50000, check_termination=1, polish_refine_iter=20, **kw)


def qp_solve(d, A, B, G, Gf, Q, R, Qf, q, l, u, settings=None):
    s = settings or default_settings()
    A, B, G, Gf, Q, R, Qf, q, l, u = map(_c, (A, B, G, Gf, Q, R, Qf, q, l, u))
    nz = d.nx + d.nu
    n = nz * d.N + d.nx
    m = d.N * (d.nx + d.ni) + d.ni_f + d.nx
    assert q.size == n and l.size == m and u.size == m
    x = np.zeros(n)
    y = np.zeros(m)
    info = OsqpInfo()
    lib().so_qp_solve(C.byref(d), _p(A), _p(B), _p(G), _p(Gf), _p(Q), _p(R), _p(Qf), _p(q), _p(l), _p(u), C.byref(s), _p(x), _p(y), C.byref(info))
    return x, y, info


def qp_kkt(d, A, B, G, Gf, Q, R, Qf, q, l, u, x, y):
    A, B, G, Gf, Q, R, Qf, q, l, u, x, y = map(_c, (A, B, G, Gf, Q, R, Qf, q, l, u, x, y))
    out = np.zeros(4)
    lib().so_qp_kkt(C.byref(d), _p(A), _p(B), _p(G), _p(Gf), _p(Q), _p(R), _p(Qf), _p(q), _p(l), _p(u), _p(x), _p(y), _p(out))
    return dict(stationarity=out[0], primal=out[1], dual_sign=out[2], complementarity=out[3])


class OracleQP:
    """Bounds / cost bookkeeping of the reference's QP class (single instance)."""

    def __init__(self, d, G, Gf, g, gf, Q, R, Qf, settings=None):
        self.d, self.G, self.Gf = d, _c(G), _c(Gf)
        self.g_model, self.gf_model = _c(g), _c(gf)
        self.Q, self.R, self.Qf = _c(Q), _c(R), _c(Qf)
        self.settings = settings
        nx, nu, N, ni, nif = d.nx, d.nu, d.N, d.ni, d.ni_f
        self.n = (nx + nu) * N + nx
        self.mb = N * (nx + ni) + nif  # rows without the x0 pin
        # lbg nominal: 0 on dynamics rows, -inf on inequalities (qp_jit.py:154-156)
        lb = np.tile(np.concatenate([np.zeros(nx), -np.inf * np.ones(ni)]), N)
        self.lbg_nominal = np.concatenate([lb, -np.inf * np.ones(nif)])
        self.A = self.B = None
        self.q = np.zeros(self.n)
        self.ubg = None
        self.lbg = self.lbg_nominal.copy()
        self.last_info = None

    def update_dynamics(self, A, B, g_list):
        """qp_jit.py:518-576 with _reassemble_numeric_same_sparsity :194-273: ubg <- [0; g_k]..., g_N ; lbg reset."""
        d = self.d
        self.A, self.B = _c(A), _c(B)
        pieces = [np.concatenate([np.zeros(d.nx), np.asarray(g_list[k], dtype=float).ravel()]) for k in range(d.N)]
        self.ubg = np.concatenate(pieces + [np.asarray(g_list[d.N], dtype=float).ravel()])
        self.lbg = self.lbg_nominal.copy()

    def offset_constraints(self, c):
        """qp_jit.py:595-610: c is (nx, N).  +EPSILON lands on ALL rows of ubg (quirk q3)."""
        d = self.d
        assert c.shape == (d.nx, d.N)
        off = np.concatenate([np.concatenate([c[:, k], np.zeros(d.ni)]) for k in range(d.N)] + [np.zeros(d.ni_f)])
        self.ubg = self.ubg - off + EPSILON
        self.lbg = self.lbg_nominal - off - EPSILON

    def update_ubg(self, ubg):
        self.ubg = np.asarray(ubg, dtype=float).copy()

    def update_q_cost_lin(self, q):
        q = np.asarray(q, dtype=float).ravel()
        assert q.size == self.n
        self.q = q.copy()

    def add_q_cost_lin(self, q):
        self.q = self.q + np.asarray(q, dtype=float).ravel()

    def bounds_with_x0(self, x0):
        """qp_jit.py:370-386."""
        x0 = np.asarray(x0, dtype=float).ravel()
        l = np.concatenate([self.lbg, -x0 - EPSILON])
        u = np.concatenate([self.ubg, -x0 + EPSILON])
        l = np.where(np.isneginf(l), -BIG, l)
        l = np.where(np.isposinf(l), BIG, l)
        u = np.where(np.isposinf(u), BIG, u)
        u = np.where(np.isneginf(u), -BIG, u)
        return l, u

    def solve(self, x0):
        d = self.d
        l, u = self.bounds_with_x0(x0)
        if getattr(self, "backend", None) is not None:
            # alternative exact solver supplied by the tests (dense interior point, tests/problems.py:ipm_backend) for QPs on which the ADMM
            # restatement does not reach eps 1e-9 within its iteration cap; same data, same row layout, same packing
            x, y, info = self.backend(self, l, u)
        else:
            x, y, info = qp_solve(d, self.A, self.B, self.G, self.Gf, self.Q, self.R, self.Qf, self.q, l, u, self.settings)
        self.last_info = info
        if info.status not in (1, 2):  # qp_jit.py:397
            return {"success": False, "status": info.status}
        return self.pack_solution(x, y, info.obj_val, info.setup_time_ms + info.solve_time_ms)

    def pack_solution(self, x, y, cost, time_ms=None):
        """qp_jit.py:487-513."""
        d = self.d
        nx, nu, N, ni, nif = d.nx, d.nu, d.N, d.ni, d.ni_f
        nz = nx + nu
        X = np.zeros((nx, N + 1))
        U = np.zeros((nu, N))
        for k in range(N):
            X[:, k] = x[k * nz: k * nz + nx]
            U[:, k] = x[k * nz + nx: (k + 1) * nz]
        X[:, N] = x[N * nz:]
        lam = y[:-nx]
        mu_f = lam[-nif:]
        mu = lam[:-nif].reshape(N, nx + ni)[:, nx:].T
        return dict(success=True, primal_vec=x.copy(), primal_x=X, primal_u=U, dual_vec=lam.copy(), dual_mu=mu.copy(),
                    dual_mu_f=mu_f.copy(), cost=float(cost), time_ms=time_ms)


class OracleFastSLS:
    """Single-instance restatement of fast_SLS (solver/fast_SLS_jit.py:195-646), quirks q2-q5 included."""

    def __init__(self, d, G, Gf, g, gf, E, Q, R, Qf, Q_reg, R_reg, Q_reg_f, settings=None):
        self.d = d
        self.G, self.Gf, self.gf_raw = _c(G), _c(Gf), _c(gf)
        self.Q_reg, self.R_reg, self.Q_reg_f = _c(Q_reg), _c(R_reg), _c(Q_reg_f)
        self.qp = OracleQP(d, G, Gf, g, gf, Q, R, Qf, settings)
        self.eps = 1e-10  # epsilon_backoff :205
        self.MAX_ITER = 30  # :206
        self.rti_steps = None
        self.E_default = _c(E)
        self._prev_primal = None  # persists across solve() calls and resets (quirk q5)
        self.cur = {}
        self.initialize_solver()
        self.initialize_backoff()

    def set_rti_steps(self, steps):
        self.rti_steps = None if (steps is None or steps <= 0) else int(steps)

    # :408-454
    def initialize_solver(self):
        d = self.d
        self.cur = {"eta": np.zeros((d.N, d.N, d.ni)), "eta_f": np.zeros((d.N + 1, d.ni_f)), "iteration_number": 0,
                    "success": False}

    def initialize_backoff(self):
        d = self.d
        c = self.cur
        c["beta"] = np.full((d.N, d.N, d.ni), self.eps)
        c["beta_f"] = np.full((d.N + 1, d.ni_f), self.eps)
        c["backoff"] = np.sqrt(c["beta"]).sum(axis=1)
        c["backoff_f"] = np.sqrt(c["beta_f"]).sum(axis=0)
        c["backoff_x"] = np.zeros((d.N + 1, d.nx))
        c["backoff_u"] = np.zeros((d.N, d.nu))

    def reset_solver_to_zeros(self):  # :424-442
        self.initialize_solver()
        self.initialize_backoff()
        self.qp.ubg = None
        self.qp.lbg = self.qp.lbg_nominal.copy()
        self.qp.q = np.zeros(self.qp.n)

    def update_dynamics_list(self, A, B, E=None, g_list=None, c_list=None):  # :250-273
        d = self.d
        self.A, self.B = _c(A), _c(B)
        if E is not None:
            self.E = _c(E)
        if g_list is not None:
            self.g_list = [np.asarray(g, dtype=float).ravel() for g in g_list]
        self.qp.update_dynamics(self.A, self.B, self.g_list)
        if c_list is not None:
            self.c = _c(c_list).reshape(d.N, d.nx)
            self.qp.offset_constraints(self.c.T)

    def update_linear_cost(self, q):
        self.qp.update_q_cost_lin(q)

    def forward_solve(self, x0):  # :459-473
        sol = self.qp.solve(x0)
        if not sol["success"]:
            return False
        c = self.cur
        for k in ("primal_vec", "primal_x", "primal_u", "dual_vec", "dual_mu", "dual_mu_f"):
            c[k] = sol[k]
        c["cost_nominal"] = sol["cost"]
        return True

    def evaluate_dual_eta(self):  # :475-487
        d, c = self.d, self.cur
        beta = np.maximum(c["beta"], self.eps)
        beta_f = np.maximum(c["beta_f"], self.eps)
        for jj in range(d.N):
            for kk in range(jj, d.N):
                c["eta"][kk, jj] = c["dual_mu"][:, kk] / (2.0 * np.sqrt(beta[kk, jj]))
        for jj in range(d.N + 1):
            c["eta_f"][jj] = c["dual_mu_f"] / (2.0 * np.sqrt(beta_f[jj]))

    def check_convergence(self):  # :581-600 (only the primal test is returned)
        p = self.cur["primal_vec"]
        if self._prev_primal is None:
            self._prev_primal = p.copy()
            return False
        ok = np.max(np.abs(p - self._prev_primal)) <= 1e-3
        self._prev_primal = p.copy()
        return bool(ok)

    def backward_and_tighten(self):  # :489-571
        d, c = self.d, self.cur
        S, K = backward(d, self.A, self.B, self.G, self.Gf, c["eta"], c["eta_f"], self.Q_reg, self.R_reg, self.Q_reg_f)
        c["K"] = K
        Px, Pu = propagate(d, self.A, self.B, self.E, K)
        beta, beta_f, bo, bof = backoff(d, Px, Pu, self.G, self.Gf, self.eps)
        c.update(beta=beta, beta_f=beta_f, backoff=bo, backoff_f=bof)
        c["backoff_x"] = np.vstack((bo[:, : d.nx], bof[: d.nx]))  # :557 (quirk q4)
        c["backoff_u"] = bo[:, d.nx: d.nx + d.nu]
        g = np.stack(self.g_list[:-1])  # (N, ni)
        tab = np.vstack([-self.c.T, (g - bo).T])  # :566
        new_ubg = np.concatenate([tab.reshape(-1, order="F"), self.gf_raw - bof])  # :567-568 raw gf (quirk q2)
        self.qp.update_ubg(new_ubg)  # no +EPSILON (quirk q3)

    def _step(self, x0):  # :314-327
        if not self.forward_solve(x0):
            return False
        self.evaluate_dual_eta()
        if self.check_convergence():
            self.cur["success"] = True
            return True
        self.backward_and_tighten()
        self.cur["iteration_number"] += 1
        return None

    def solve(self, x0):  # :278-312
        if self.rti_steps is not None and self.rti_steps > 0:
            self.initialize_backoff()
            last_infeasible = False
            for _ in range(self.rti_steps):
                if self._step(x0) is False:
                    last_infeasible = True
                    break
            if not last_infeasible:
                self.forward_solve(x0)
            self.cur["success"] = (not last_infeasible) or bool(self.cur.get("success", False))
            return self.post()
        self.initialize_backoff()
        for i in range(self.MAX_ITER):
            st = self._step(x0)
            if st is False:
                self.cur["success"] = False
                out = self.post()
                self.reset_solver_to_zeros()
                return out
            if st is True:
                return self.post()
        self.forward_solve(x0)
        self.cur["success"] = False
        out = self.post()
        self.reset_solver_to_zeros()
        return out

    def post(self):  # :602-646
        c = self.cur
        keys = ("iteration_number", "success", "cost_nominal", "primal_x", "primal_u", "primal_vec", "dual_vec", "dual_mu",
                "dual_mu_f", "eta", "eta_f", "K", "beta", "beta_f", "backoff", "backoff_f", "backoff_x", "backoff_u")
        out = {k: (c[k].copy() if isinstance(c.get(k), np.ndarray) else c.get(k)) for k in keys}
        out["cost_tube"] = np.nan
        out["cost"] = np.nan
        return out


def rti_step_batch(d, A, B, g, gN, c, q, x0, G, Gf, gf_raw, E, Q, R, Qf, Q_reg, R_reg, Q_reg_f, settings=None, nthreads=1, budget_s=0.0):
    """One RTI fast-SLS step (rti_steps = 1: QP, eta, sweep, tightened QP) for a batch of fresh instances, driven by C threads
    (so_rti_step_batch): the CPU baseline of bench.py without the interpreter lock.  Arrays carry a leading batch axis.
    Returns (primal (nb,n), ok (nb) int32, done): instances are processed in order until all are done or budget_s seconds have passed."""
    s = settings or default_settings()
    A, B, g, gN, c, q, x0 = map(_c, (A, B, g, gN, c, q, x0))
    G, Gf, gf_raw, E, Q, R, Qf, Q_reg, R_reg, Q_reg_f = map(_c, (G, Gf, gf_raw, E, Q, R, Qf, Q_reg, R_reg, Q_reg_f))
    nb = A.shape[0]
    n = (d.nx + d.nu) * d.N + d.nx
    primal = np.zeros((nb, n))
    ok = np.zeros(nb, dtype=np.int32)
    f = lib().so_rti_step_batch
    f.restype = C.c_int
    done = f(C.byref(d), C.c_int(nb), C.c_int(int(nthreads)), C.c_double(float(budget_s)), _p(A), _p(B), _p(g), _p(gN), _p(c), _p(q), _p(x0),
             _p(G), _p(Gf), _p(gf_raw), _p(E), _p(Q), _p(R), _p(Qf), _p(Q_reg), _p(R_reg), _p(Q_reg_f), C.byref(s), _p(primal),
             ok.ctypes.data_as(C.c_void_p))
    return primal, ok, int(done)
