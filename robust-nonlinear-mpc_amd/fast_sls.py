"""Host-side mirror of the reference's fast-SLS interface, batched, on top of the C-ABI (include/slsqp.h).

`BatchedFastSLS` keeps the method names, argument meaning and result keys of the reference's
`fast_SLS` (solver/fast_SLS_jit.py:195-646) with a leading batch axis B on every array:

    reference (single instance)                      here (B instances, one GPU)
    fast_SLS(N,Q,R,m_LTV,Qf,Q_reg,R_reg,Q_reg_f)     BatchedFastSLS(N,Q,R,model,Qf,Q_reg,R_reg,Q_reg_f,batch=B)
    .update_dynamics_list(A,B,E,g,c)                 .update_dynamics_list(A[B,N,nx,nx],B[B,N,nx,nu],E[N+1,nx,nw],
                                                                           g[B,N,ni],g_N[B,ni_f],c[B,N,nx])
    .update_linear_cost(q)                           .update_linear_cost(q[B,n])
    .set_rti_steps(k)                                .set_rti_steps(k)
    .solve(x0) -> dict                               .solve(x0[B,nx]) -> dict of [B,...] arrays (same keys)
    .reset_solver_to_zeros()                         .reset_solver_to_zeros()

`fast_SLS` (lower case, the reference's class name) is the B=1 view returning the reference's exact shapes,
so SCP_SLS-style callers can switch by changing one import.  All arithmetic runs in the HIP library; this
module only moves arrays.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _ModelView:
    """What the path needs from the plant / LTV object, as numpy.  Accepts this package's ModelData as well as the reference's
    `LTV(m, N)` (dyn/LTV.py:17-32: nx, nu, nw, ni, ni_f, G, Gf, gf and nothing else; G may be a casadi DM)."""

    def __init__(self, model):
        self.src = model
        self.nx, self.nu, self.nw = int(model.nx), int(model.nu), int(model.nw)
        self.nz = self.nx + self.nu
        self.ni, self.ni_f = int(model.ni), int(model.ni_f)
        self.G, self.Gf = _c(np.asarray(model.G, dtype=float)), _c(np.asarray(model.Gf, dtype=float))
        self.gf = _c(np.asarray(model.gf, dtype=float)).ravel()
        E = getattr(model, "E", None)            # LTV starts from all-ones blocks and gets E through update_dynamics_list
        self.E = np.ones((self.nx, self.nw)) if E is None else _c(np.asarray(E, dtype=float))
        self.model_id = getattr(model, "model_id", None)
        g = getattr(model, "g", None)
        self.g = None if g is None else _c(np.asarray(g, dtype=float)).ravel()

    def __getattr__(self, k):                     # weights, reference points, ... of ModelData
        return getattr(self.src, k)


class _SolverForward:
    """The attributes SCP_SLS and the closed-loop scripts touch on `fast_SLS.solver_forward` (the reference's QP object):
    P_mat_csc (SCP_SLS_jit.py:389), ubg / update_ubg (:86-99, :521), verbose / export_standard_qp (expe/main_*:86-91)."""

    def __init__(self, owner):
        self._o = owner
        self.verbose = False
        self.export_standard_qp = False

    @property
    def P_mat_csc(self):
        from scipy.sparse import block_diag, csc_matrix
        o = self._o
        return csc_matrix(block_diag([o.Q, o.R] * o.N + [o.Qf]))      # P = blkdiag(Q,R,...,Qf) (qp_jit.py:126-130); OSQP gets 2P

    @property
    def ubg(self):
        u = self._o.get("ubg", (self._o.mb,))
        return u[0] if self._o.B == 1 else u

    @property
    def lbg(self):
        l = self._o.get("lbg", (self._o.mb,))
        return l[0] if self._o.B == 1 else l

    def update_ubg(self, ubg):
        o = self._o
        u = _c(np.asarray(ubg, dtype=float)).reshape(o.B, o.mb)
        L.check(o.lib.slsqp_set(o.h, b"ubg", _ptr(u), L.HOST))


class BatchedFastSLS:
    def __init__(self, N, Q, R, model, Qf, Q_reg=None, R_reg=None, Q_reg_f=None, batch=1, device=0):
        self.lib = L.load()
        self.h = None
        model = model if isinstance(model, _ModelView) else _ModelView(model)
        self.N, self.m, self.B = int(N), model, int(batch)
        nx, nu = model.nx, model.nu
        self.Q, self.R, self.Qf = _c(Q), _c(R), _c(Qf)
        # OCP defaults (solver/ocp.py:14-27)
        self._Q_reg = _c(np.eye(nx) if Q_reg is None else Q_reg)
        self._R_reg = _c(np.eye(nu) if R_reg is None else R_reg)
        self._Q_reg_f = _c(np.eye(nx) if Q_reg_f is None else Q_reg_f)
        self.solver_forward = _SolverForward(self)
        self.dims = L.Dims(nx, nu, model.nw, self.N, model.ni, model.ni_f)
        self.n = model.nz * self.N + nx
        self.mb = self.N * (nx + model.ni) + model.ni_f
        self.h = self.lib.slsqp_create(C.byref(self.dims), self.B, device)
        if not self.h:
            raise RuntimeError("slsqp_create: " + self.lib.slsqp_last_error().decode())
        self.opts = L.Opts()
        self.lib.slsqp_default_opts(C.byref(self.opts))
        self.opts.rti_steps = 0       # fast_SLS default: iterate to convergence (fast_SLS_jit.py:214)
        import os
        self.opts.precision = int(os.environ.get("SLSQP_PRECISION", "0"))   # 0 fp64, 1 mixed fp32/fp64 (see include/slsqp.h)
        self.verbose = False
        self.save_it_data = False
        self.CONV_EPS = 1e-6
        self._push_costs()
        G, Gf, gf = _c(model.G), _c(model.Gf), _c(model.gf)
        L.check(self.lib.slsqp_set_constraints(self.h, _ptr(G), _ptr(Gf), _ptr(gf)))
        self._E = _c(np.stack([model.E] * (self.N + 1)))
        if model.model_id is not None and model.g is not None:
            g_raw = _c(model.g)
            L.check(self.lib.slsqp_set_model(self.h, int(model.model_id), _ptr(g_raw)))
            L.check(self.lib.slsqp_set_E(self.h, _ptr(self._E), L.HOST))

    def _push_costs(self):
        L.check(self.lib.slsqp_set_costs(self.h, _ptr(self.Q), _ptr(self.R), _ptr(self.Qf), _ptr(self._Q_reg), _ptr(self._R_reg), _ptr(self._Q_reg_f)))

    # SCP_SLS assigns these after construction (`self.fast_SLS_solver.Q_reg = self.Q_reg`, SCP_SLS_jit.py:386-388): assignment takes effect
    def _reg_prop(name):
        def get(self):
            return getattr(self, name)

        def put(self, v):
            setattr(self, name, _c(np.asarray(v, dtype=float)))
            if getattr(self, "h", None):
                self._push_costs()
        return property(get, put)

    Q_reg, R_reg, Q_reg_f = _reg_prop("_Q_reg"), _reg_prop("_R_reg"), _reg_prop("_Q_reg_f")
    del _reg_prop

    def close(self):
        if getattr(self, "h", None):
            self.lib.slsqp_destroy(self.h)
            self.h = None

    def __del__(self):
        # not at interpreter shutdown: the HIP runtime may already be gone then, and calling into it aborts the process
        import sys
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    # ---- reference surface -------------------------------------------------------------------------
    def set_rti_steps(self, steps):
        self.opts.rti_steps = 0 if (steps is None or steps <= 0) else int(steps)

    def set_regularisers(self, Q_reg, R_reg, Q_reg_f):
        """All three at once (one push)."""
        self._Q_reg, self._R_reg, self._Q_reg_f = _c(Q_reg), _c(R_reg), _c(Q_reg_f)
        self._push_costs()

    def update_dynamics_list(self, A, Bm, E=None, g=None, g_N=None, c=None):
        B, N, m = self.B, self.N, self.m
        A, Bm, g, g_N, c = _c(A), _c(Bm), _c(g), _c(g_N), _c(c)
        assert A.shape == (B, N, m.nx, m.nx) and Bm.shape == (B, N, m.nx, m.nu), (A.shape, Bm.shape)
        assert g.shape == (B, N, m.ni) and g_N.shape == (B, m.ni_f) and c.shape == (B, N, m.nx)
        if E is not None:
            self._E = _c(E)
            assert self._E.shape == (N + 1, m.nx, m.nw)
        L.check(self.lib.slsqp_update_dynamics(self.h, _ptr(A), _ptr(Bm), _ptr(self._E), _ptr(g), _ptr(g_N), _ptr(c), L.HOST))

    def linearize(self, X, U):
        """Batched SCP_SLS.update_jacobian on the GPU: X (B,N+1,nx), U (B,N,nu) nominal trajectories -> A,B,c,g,q and the
        un-tightened bounds, all left on the device (SCP_SLS_jit.py:251-366)."""
        X, U = _c(X), _c(U)
        assert X.shape == (self.B, self.N + 1, self.m.nx) and U.shape == (self.B, self.N, self.m.nu)
        L.check(self.lib.slsqp_linearize(self.h, _ptr(X), _ptr(U), L.HOST))

    def update_linear_cost(self, q):
        q = _c(q)
        assert q.shape == (self.B, self.n)
        L.check(self.lib.slsqp_update_linear_cost(self.h, _ptr(q), L.HOST))

    def add_linear_cost(self, q):
        """fast_SLS.add_linear_cost (fast_SLS_jit.py:578-579): q_lin += q."""
        q = _c(q).reshape(self.B, self.n)
        self.update_linear_cost(self.get("q", (self.n,)) + q)

    def reset_solver_to_zeros(self):
        L.check(self.lib.slsqp_reset(self.h))

    def get(self, name, shape, dtype=np.float64):
        out = np.empty((self.B,) + tuple(shape), dtype=dtype)
        want = self.lib.slsqp_result_bytes(self.h, name.encode())
        if want >= 0 and want * self.B != out.nbytes:      # the library copies `want` bytes per instance: a smaller buffer would be overrun
            raise ValueError(f"result {name!r}: the library holds {want} bytes per instance, shape {tuple(shape)} of {np.dtype(dtype)} is {out.nbytes // max(1, self.B)}")
        L.check(self.lib.slsqp_get(self.h, name.encode(), out.ctypes.data_as(C.c_void_p), L.HOST))
        return out

    def timing_ms(self):
        t = np.zeros(L.TIMING_LEN)
        L.check(self.lib.slsqp_last_timing(self.h, _ptr(t), t.size))
        return dict(total=t[0], qp=t[1], sweep=t[2], other=t[3], jac=t[4])

    def kernel_timing(self):
        """(total ms, launches) of the dominant QP kernel since the last call (HIP events on the handle's stream; needs opts.time_kernels = 1);
        the device counters of the work those launches did land in attributes."""
        t = np.zeros(L.KERNEL_TIMING_LEN)
        L.check(self.lib.slsqp_kernel_timing(self.h, _ptr(t), t.size))
        self.mx_retries = int(t[2])            # instances re-solved in fp64 after a mixed-precision attempt (opts.precision = 1)
        self.fwd_instance_sweeps = int(t[3])   # instance forward sweeps (= backward sweeps)
        self.fwd_factor_sweeps = int(t[4])     # ... of which factorising
        self.factor_stages = int(t[5])         # stages factorised
        self.qp_solves = int(t[6])             # QP solves that ran (at least one block solve) in k_qp_solve launches
        self.bwd_sweeps_skipped = int(t[7])    # block solves that ended after the forward sweep (residual check of an already certified solve)
        return t[0], int(t[1])

    def solve(self, x0, fetch=True):
        """x0: (B,nx) = x_nom0 - x_meas (the argument SCP_SLS.socp_step passes, SCP_SLS_jit.py:408-410)."""
        x0 = _c(x0)
        assert x0.shape == (self.B, self.m.nx)
        L.check(self.lib.slsqp_solve(self.h, _ptr(x0), L.HOST, C.byref(self.opts)))
        return self.post_processing_solution() if fetch else None

    def post_processing_solution(self):
        """Keys of fast_SLS.post_processing_solution (fast_SLS_jit.py:615-643), batch axis first."""
        m, N, B = self.m, self.N, self.B
        nx, nu, ni, nif, nz = m.nx, m.nu, m.ni, m.ni_f, m.nz
        pv = self.get("primal_vec", (self.n,))
        dv = self.get("dual_vec", (self.mb,))
        st = pv[:, : nz * N].reshape(B, N, nz)
        primal_x = np.concatenate([st[:, :, :nx], pv[:, None, nz * N:]], axis=1).transpose(0, 2, 1).copy()  # (B,nx,N+1)
        primal_u = st[:, :, nx:].transpose(0, 2, 1).copy()                                                  # (B,nu,N)
        dual_mu_f = dv[:, -nif:].copy()
        dual_mu = dv[:, :-nif].reshape(B, N, nx + ni)[:, :, nx:].transpose(0, 2, 1).copy()                   # (B,ni,N)
        K = self.get("K", (N, N + 1, nu, nx))
        out = dict(
            iteration_number=self.get("iteration_number", (), np.int32),
            success=self.get("success", (), np.int32).astype(bool),
            status=self.get("status", (), np.int32),
            qp_iters=self.get("qp_iters", (), np.int32),
            cost_nominal=self.get("cost_nominal", ()),
            cost_tube=np.full(B, np.nan), cost=np.full(B, np.nan),     # NaN in the reference's dict too (fast_SLS_jit.py:619-620)
            cost_tube_value=self.get("cost_tube", ()),                # what the reference computes and prints (:540-544, SLS.eval_cost)
            primal_x=primal_x, primal_u=primal_u, primal_vec=pv, dual_vec=dv, dual_mu=dual_mu, dual_mu_f=dual_mu_f,
            eta=self.get("eta", (N, N, ni)), eta_f=self.get("eta_f", (N + 1, nif)),
            K=K, K_mat=K.transpose(0, 1, 3, 2, 4).reshape(B, N * nu, (N + 1) * nx),
            Phi_x=None, Phi_u=None, Phi_x_mat=None, Phi_u_mat=None,
            beta=self.get("beta", (N, N, ni)), beta_f=self.get("beta_f", (N + 1, nif)),
            backoff=self.get("backoff", (N, ni)), backoff_f=self.get("backoff_f", (nif,)),
            backoff_x=self.get("backoff_x", (N + 1, nx)), backoff_u=self.get("backoff_u", (N, nu)),
            kkt=self.get("kkt", (8,)),
        )
        t = self.timing_ms()
        out["t_qp_ms"] = t["qp"]
        out["t_backward_ms"] = t["sweep"]
        return out

    # ---- QP-level / sweep-level boundaries (tests, drop-in for the generated-OSQP module) -------------
    def qp_nnz(self):
        n, m, nP, nA = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.lib.slsqp_qp_nnz(C.byref(self.dims), C.byref(n), C.byref(m), C.byref(nP), C.byref(nA))
        return n.value, m.value, nP.value, nA.value

    def qp_update_data_mat(self, P_x=None, A_x=None):
        P_x = None if P_x is None else _c(P_x)
        A_x = None if A_x is None else _c(A_x)
        L.check(self.lib.slsqp_qp_update_data_mat(self.h, _ptr(P_x), _ptr(A_x), L.HOST))
        return 0

    def qp_update_data_vec(self, q=None, l=None, u=None):
        q, l, u = (None if v is None else _c(v) for v in (q, l, u))
        L.check(self.lib.slsqp_qp_update_data_vec(self.h, _ptr(q), _ptr(l), _ptr(u), L.HOST))
        return 0

    def qp_solve(self):
        n, m = self.n, self.mb + self.m.nx
        x = np.empty((self.B, n))
        y = np.empty((self.B, m))
        st = np.empty(self.B, dtype=np.int32)
        it = np.empty(self.B, dtype=np.int32)
        L.check(self.lib.slsqp_qp_solve(self.h, _ptr(x), _ptr(y), st.ctypes.data_as(C.c_void_p), it.ctypes.data_as(C.c_void_p), L.HOST, C.byref(self.opts)))
        return x, y, st, it, self.timing_ms()["qp"] * 1e-3

    def sweep(self, eta, eta_f):
        m, N, B = self.m, self.N, self.B
        eta, eta_f = _c(eta), _c(eta_f)
        K = np.empty((B, N, N + 1, m.nu, m.nx))
        beta = np.empty((B, N, N, m.ni))
        beta_f = np.empty((B, N + 1, m.ni_f))
        bo = np.empty((B, N, m.ni))
        bof = np.empty((B, m.ni_f))
        L.check(self.lib.slsqp_sweep(self.h, _ptr(eta), _ptr(eta_f), _ptr(K), _ptr(beta), _ptr(beta_f), _ptr(bo), _ptr(bof), L.HOST))
        return dict(K=K, beta=beta, beta_f=beta_f, backoff=bo, backoff_f=bof, cost_tube_value=self.get("cost_tube", ()))


class fast_SLS(BatchedFastSLS):
    """B = 1 view with the reference's exact argument and result shapes (fast_SLS_jit.py:195)."""

    def __init__(self, N, Q, R, m, Qf, Q_reg=None, R_reg=None, Q_reg_f=None, device=0):
        super().__init__(N, Q, R, m, Qf, Q_reg, R_reg, Q_reg_f, batch=1, device=device)

    def update_dynamics_list(self, new_list_A, new_list_B, new_list_E=None, new_list_g=None, c_offset_list=None):
        A = np.stack([np.asarray(a, dtype=float) for a in new_list_A])[None]
        Bm = np.stack([np.asarray(b, dtype=float) for b in new_list_B])[None]
        E = None if new_list_E is None else np.stack([np.asarray(e, dtype=float) for e in new_list_E])
        g = np.stack([np.asarray(x, dtype=float).ravel() for x in new_list_g[:-1]])[None]
        gN = np.asarray(new_list_g[-1], dtype=float).ravel()[None]
        c = np.stack([np.asarray(x, dtype=float).ravel() for x in c_offset_list])[None]
        super().update_dynamics_list(A, Bm, E, g, gN, c)

    def update_linear_cost(self, q_cost_lin):
        super().update_linear_cost(np.asarray(q_cost_lin, dtype=float).reshape(1, -1))

    def add_linear_cost(self, q_cost_lin):
        super().add_linear_cost(np.asarray(q_cost_lin, dtype=float).reshape(1, -1))

    def solve(self, x0):
        out = super().solve(np.asarray(x0, dtype=float).reshape(1, -1))
        return {k: (v[0] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 1 else v) for k, v in out.items()}


class DeviceBatch:
    """Device-resident inputs of one batch (torch CUDA tensors used purely as HBM buffers) so that repeated MPC steps do
    not cross PCIe: update_dynamics / update_linear_cost / solve are called with SLSQP_DEVICE pointers."""

    def __init__(self, solver, batch):
        import torch
        self.torch = torch
        self.f = solver
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        self.A, self.Bm, self.E = t(batch["A"]), t(batch["B"]), t(batch["E"])
        self.g, self.gN, self.c, self.q, self.x0 = t(batch["g"]), t(batch["gN"]), t(batch["c"]), t(batch["q"]), t(batch["x0_arg"])
        self.x0_alt = (-self.x0).contiguous()
        self.nstep = 0
        torch.cuda.synchronize()

    def step(self):
        """One MPC step of the hot path: update_dynamics_list + update_linear_cost + solve (RTI: 2 QPs + 1 sweep).
        The measured state alternates between two values so that consecutive steps differ by more than the 1e-3 of
        check_convergence_socp: otherwise the reference's leaked convergence state would skip the
        Riccati sweep and tightening on every step after the first, and the step would not be the full hot path."""
        f, p = self.f, (lambda x: C.c_void_p(x.data_ptr()))
        x0 = self.x0 if self.nstep % 2 == 0 else self.x0_alt
        self.nstep += 1
        L.check(f.lib.slsqp_update_dynamics(f.h, p(self.A), p(self.Bm), p(self.E), p(self.g), p(self.gN), p(self.c), L.DEVICE))
        L.check(f.lib.slsqp_update_linear_cost(f.h, p(self.q), L.DEVICE))
        L.check(f.lib.slsqp_solve(f.h, p(x0), L.DEVICE, C.byref(f.opts)))

    def fetch_device(self, name, shape):
        out = self.torch.empty((self.f.B,) + tuple(shape), dtype=self.torch.float64, device=self.A.device)
        L.check(self.f.lib.slsqp_get(self.f.h, name.encode(), C.c_void_p(out.data_ptr()), L.DEVICE))
        return out


class SlicedDeviceBatch:
    """One batch cut into K contiguous slices, each with its own handle (= its own HIP stream) and its own host thread.

    MPC instances are independent, so the slices need not advance in lockstep: `run(steps)` lets every slice do its `steps` MPC
    steps back to back on its own thread, and the few-instance tails of one slice's QP solves (latency-bound launches that leave
    the GPU almost idle) overlap the bulk launches of the others (DESIGN.md section 6 holds the measured slice counts).  The C-ABI
    calls release the GIL."""

    KEYS = ("A", "B", "g", "gN", "c", "q", "x0_arg")

    def __init__(self, make_solver, batch, n_slices):
        B = batch["A"].shape[0]
        K = max(1, min(int(n_slices), B))
        self.bounds = [(B * k // K, B * (k + 1) // K) for k in range(K)]
        self.solvers, self.slices = [], []
        for lo, hi in self.bounds:
            sub = dict(batch)
            for key in self.KEYS:
                sub[key] = batch[key][lo:hi]
            f = make_solver(hi - lo)
            self.solvers.append(f)
            self.slices.append(DeviceBatch(f, sub))

    def run(self, steps):
        """`steps` MPC steps of every slice; returns per-slice sums of the GPU times (ms) of the QP solves, the sweeps and the whole calls."""
        import threading
        acc = [dict(qp=0.0, sweep=0.0, total=0.0) for _ in self.slices]
        err = []

        def work(k):
            try:
                d, f = self.slices[k], self.solvers[k]
                for _ in range(steps):
                    d.step()
                    t = f.timing_ms()
                    for key in acc[k]:
                        acc[k][key] += t[key]
            except Exception as e:      # surface worker failures in the caller
                err.append(e)

        if len(self.slices) == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(k,)) for k in range(len(self.slices))]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if err:
            raise err[0]
        return acc

    def step(self):
        return self.run(1)

    def fetch_device(self, name, shape):
        return self.slices[0].torch.cat([d.fetch_device(name, shape) for d in self.slices], dim=0)

    def get(self, name, shape, dtype=np.float64):
        return np.concatenate([f.get(name, shape, dtype) for f in self.solvers], axis=0)

    def kernel_timing(self):
        """Summed over the slices: (total ms of k_qp_solve launches, launches, instance sweeps, fp64 re-solves)."""
        tot = [0.0, 0, 0, 0]
        self.fwd_factor_sweeps = self.factor_stages = self.qp_solves = 0
        for f in self.solvers:
            ms, n = f.kernel_timing()
            tot[0] += ms; tot[1] += n; tot[2] += f.fwd_instance_sweeps; tot[3] += f.mx_retries
            self.fwd_factor_sweeps += f.fwd_factor_sweeps; self.factor_stages += f.factor_stages; self.qp_solves += f.qp_solves
        return tuple(tot)

    def close(self):
        for f in self.solvers:
            f.close()
