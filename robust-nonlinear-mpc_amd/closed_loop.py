"""Batched closed-loop robust MPC driver on the GPU.

Mirrors what the reference's closed-loop scripts do around `SCP_SLS` for ONE instance
(expe/main_rocket_robust_closed_loop.py:128-206, main_pendulum...:62-121, main_quadrotor...:98-158), for B instances at once:

    for i in range(steps):
        if i > 0: solver.reset_warm_start()            SCP_SLS_jit.py:500-551   (shift + solver reset)
        solution = solver.solve(x0)                     SCP_SLS_jit.py:65-152    (rti x [linearise, fast-SLS, nominal += delta])
        u0 = solution['primal_u'][:, 0]
        x0 = m.ddyn(x0, u0) + m.E @ w_i                 plant + bounded noise (rocket only)

Everything between `reset(...)` and the result arrays stays on the device (slsqp_cl_step); the host only supplies the
disturbance samples.  The reference obtains the very first nominal trajectory from IPOPT (out of scope): pass it as
`X_nom, U_nom`, or let the driver roll the plant out from x0 under a constant input.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .fast_sls import BatchedFastSLS, _c, _ptr


class ClosedLoopMPC:
    def __init__(self, model, N, batch, rti=None, fast_sls_rti_steps=None, device=0):
        m = model
        self.m, self.N, self.B = m, int(N), int(batch)
        self.rti = int(m.rti if rti is None else rti)
        self.f = BatchedFastSLS(self.N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=self.B, device=device)
        # rti > 0: that many SCP iterations per MPC step (scripts); rti <= 0: SCP_SLS's default converge mode (SCP_SLS_jit.py:20-21).
        # fast_sls_rti_steps: None -> the script's value when rti is the script's, the reference default (converge, None) otherwise
        if fast_sls_rti_steps is None and rti is None:
            fast_sls_rti_steps = m.fast_sls_rti_steps
        self.f.set_rti_steps(fast_sls_rti_steps)
        self.steps_done = 0

    def close(self):
        self.f.close()

    def reset(self, x_meas, X_nom=None, U_nom=None, u_init=None, solve_nominal=False, max_qp=120, tol=1e-7, rho=1e3, continuation=1):
        """x_meas (B,nx).  X_nom (B,N+1,nx), U_nom (B,N,nu) optional initial nominal; else roll-out under `u_init` (default: the
        model's neutral input).  solve_nominal=True then solves the nominal NLP from that guess on the GPU (the role IPOPT has in
        SCP_SLS.solve_nominal_trajectory, SCP_SLS_jit.py:161-188); per-instance outcome in self.nlp_status (0 = KKT point found).
        continuation=K > 1 (far-away states): the NLP is solved for x_ref + s (x_meas - x_ref), s = 1/K, 2/K, ..., 1, each stage from the
        previous stage's trajectory (the initial-state constraint is damped like the others, so a stage moves x_0 gradually)."""
        f, m = self.f, self.m
        x_meas = _c(x_meas)
        assert x_meas.shape == (self.B, m.nx)
        Xn = None if X_nom is None else _c(X_nom)
        Un = None if U_nom is None else _c(U_nom)
        ui = _c(m.u_ref if u_init is None else u_init)
        K = max(1, int(continuation)) if solve_nominal and Xn is None else 1
        x_first = x_meas if K == 1 else _c(m.x_ref + (x_meas - m.x_ref) / K)
        L.check(f.lib.slsqp_cl_init(f.h, _ptr(x_first), _ptr(Xn), _ptr(Un), _ptr(ui), L.HOST))
        self.steps_done = 0
        self.nlp_status = None
        if solve_nominal:
            L.check(f.lib.slsqp_nominal_solve(f.h, int(max_qp), float(tol), float(rho), C.byref(f.opts)))
            for k in range(2, K + 1):
                xs = _c(m.x_ref + (x_meas - m.x_ref) * (k / K))
                L.check(f.lib.slsqp_set(f.h, b"x_meas", _ptr(xs), L.HOST))
                L.check(f.lib.slsqp_nominal_solve(f.h, int(max_qp), float(tol), float(rho), C.byref(f.opts)))
            self.nlp_status = f.get("nlp_status", (), np.int32)
            self.nlp_iterations = f.get("nlp_iterations", (), np.int32)
            self.nlp_info = f.get("nlp_info", (12,))

    def step(self, w=None, fetch=True):
        """One MPC step of the whole batch.  w (B,nx): disturbance sample in [-1,1]^nx (x+ = ddyn(x,u0) + E w), or None."""
        f = self.f
        wv = None if w is None else _c(w)
        L.check(f.lib.slsqp_cl_step(f.h, self.rti, _ptr(wv), L.HOST, C.byref(f.opts)))
        self.steps_done += 1
        if not fetch:
            return None
        m, N = self.m, self.N
        return dict(
            u0=f.get("u0", (m.nu,)), x_next=f.get("x_meas", (m.nx,)),
            nominal_x=f.get("nominal_x", (N + 1, m.nx)), nominal_u=f.get("nominal_u", (N, m.nu)),
            backoff_x=f.get("backoff_x", (N + 1, m.nx)), backoff_u=f.get("backoff_u", (N, m.nu)),
            success=f.get("scp_success", (), np.int32).astype(bool), status=f.get("status", (), np.int32),
            scp_iterations=f.get("scp_iterations", (), np.int32), primal_infeasibility=f.get("primal_infeasibility", ()),
            t_qp_ms=f.timing_ms()["qp"], t_riccati_ms=f.timing_ms()["sweep"], t_jac_ms=f.timing_ms()["jac"],
        )

    def run_on_device(self, x0, steps, W=None, X_nom=None, U_nom=None, solve_nominal=False, continuation=1):
        """Same result as run(), but the per-step records stay in device buffers (slsqp_cl_log) and are read back once at the end: the only
        host -> device traffic per MPC step is the disturbance sample (B,nx)."""
        f, m, N, B = self.f, self.m, self.N, self.B
        L.check(f.lib.slsqp_cl_log(f.h, int(steps)))
        self.reset(x0, X_nom, U_nom, solve_nominal=solve_nominal, continuation=continuation)
        t_qp, t_ric, t_jac = np.zeros((steps, 1)), np.zeros((steps, 1)), np.zeros((steps, 1))
        for i in range(steps):
            self.step(None if W is None else W[i], fetch=False)
            t = f.timing_ms()
            t_qp[i], t_ric[i], t_jac[i] = t["qp"], t["sweep"], t["jac"]
        return self._log_result(steps, t_jac, t_qp, t_ric)

    def _log_result(self, steps, t_jac, t_qp, t_ric):
        """The arrays of the device-side log (slsqp_cl_log) laid out like the reference's npz, batch axis first."""
        f, m, N = self.f, self.m, self.N
        lx = f.get("log_nominal_x", (steps, N + 1, m.nx)); lu = f.get("log_nominal_u", (steps, N, m.nu))
        lbx = f.get("log_backoff_x", (steps, N + 1, m.nx)); lbu = f.get("log_backoff_u", (steps, N, m.nu))
        u0 = f.get("log_u0", (steps, m.nu))
        return dict(
            state_trajectory=f.get("log_state", (steps, m.nx)).transpose(0, 2, 1).copy(),
            input_trajectory=u0[:, :max(steps - 1, 0)].transpose(0, 2, 1).copy(),
            nominal_trajectory_x=lx.transpose(0, 3, 2, 1).copy(), nominal_trajectory_u=lu.transpose(0, 3, 2, 1).copy(),
            backoff_trajectory_x=lbx.transpose(0, 3, 2, 1).copy(), backoff_trajectory_u=lbu.transpose(0, 3, 2, 1).copy(),
            t_jac=t_jac, t_qp=t_qp, t_riccati=t_ric,
            success=f.get("log_success", (steps,), np.int32).astype(bool), scp_iterations=f.get("log_scp_iterations", (steps,), np.int32),
            primal_infeasibility=f.get("log_primal_infeasibility", (steps,)),
        )

    def run_decoupled(self, x0, steps, W=None, X_nom=None, U_nom=None, solve_nominal=False, continuation=1, budget_ms=8.0, cut_frac=0.0):
        """Same results as run_on_device() -- bit for bit -- through slsqp_cl_run: the instances advance through their MPC steps independently (a
        chain of QP solves that is not done budget_ms after its launch started suspends itself and resumes in the next round), so nobody waits for the
        slowest instance of a step.  With opts.cl_persistent (the default) the whole loop is ONE launch: waves take instances from a device-side
        FIFO and run one MPC step each time; budget_ms / cut_frac only matter for opts.cl_persistent = 0.  Only for the rocket script's setting
        (rti = 1, one fast-SLS step, fp64).  Adds `qp_stats` (B, steps, 2, 8), `rounds` and (persistent) `loop_stats`; the t_* arrays hold the
        run's totals in their first entry."""
        f, m, N, B = self.f, self.m, self.N, self.B
        assert self.rti == 1, "slsqp_cl_run runs rti = 1 closed loops"
        L.check(f.lib.slsqp_cl_log(f.h, int(steps)))
        self.reset(x0, X_nom, U_nom, solve_nominal=solve_nominal, continuation=continuation)
        Wc = None if W is None else _c(W)
        assert Wc is None or Wc.shape == (steps, B, m.nx)
        rounds = C.c_int(0)
        L.check(f.lib.slsqp_cl_run(f.h, int(steps), _ptr(Wc), L.HOST, C.byref(f.opts), float(budget_ms), float(cut_frac), C.byref(rounds)))
        self.steps_done = steps
        t = f.timing_ms()
        t_qp, t_ric, t_jac = np.zeros((steps, 1)), np.zeros((steps, 1)), np.zeros((steps, 1))
        t_qp[0], t_ric[0], t_jac[0] = t["qp"], t["sweep"], t["jac"]
        out = self._log_result(steps, t_jac, t_qp, t_ric)
        out["qp_stats"] = f.get("log_qp_stats", (steps, 2, 8), np.int32)
        out["rounds"] = rounds.value
        if f.opts.cl_persistent:      # one persistent launch: how busy the instance queue kept the waves
            st = (C.c_double * L.CL_RUN_STATS_LEN)()
            L.check(f.lib.slsqp_cl_run_stats(f.h, st, L.CL_RUN_STATS_LEN))
            out["loop_stats"] = dict(waves=int(st[0]), busy_ms=float(st[1]), mpc_steps=int(st[2]), launch_ms=float(st[3]))
        return out

    def run(self, x0, steps, W=None, X_nom=None, U_nom=None, solve_nominal=False):
        """Closed loop of `steps` MPC steps from x0 (B,nx); W (steps,B,nx) disturbance samples or None.  Returns arrays laid out
        like the reference's npz (expe/main_rocket_robust_closed_loop.py:189-206) with a leading batch axis."""
        m, N, B = self.m, self.N, self.B
        self.reset(x0, X_nom, U_nom, solve_nominal=solve_nominal)
        out = dict(
            state_trajectory=np.zeros((B, m.nx, steps)), input_trajectory=np.zeros((B, m.nu, max(steps - 1, 0))),
            nominal_trajectory_x=np.zeros((B, m.nx, N + 1, steps)), nominal_trajectory_u=np.zeros((B, m.nu, N, steps)),
            backoff_trajectory_x=np.zeros((B, m.nx, N + 1, steps)), backoff_trajectory_u=np.zeros((B, m.nu, N, steps)),
            t_jac=np.zeros((steps, 1)), t_qp=np.zeros((steps, 1)), t_riccati=np.zeros((steps, 1)),
            success=np.zeros((B, steps), dtype=bool), scp_iterations=np.zeros((B, steps), dtype=np.int32),
            primal_infeasibility=np.full((B, steps), np.nan),
        )
        for i in range(steps):
            r = self.step(None if W is None else W[i])
            out["state_trajectory"][:, :, i] = r["nominal_x"][:, 0, :]
            if i < steps - 1:
                out["input_trajectory"][:, :, i] = r["u0"]
            out["nominal_trajectory_x"][:, :, :, i] = r["nominal_x"].transpose(0, 2, 1)
            out["nominal_trajectory_u"][:, :, :, i] = r["nominal_u"].transpose(0, 2, 1)
            out["backoff_trajectory_x"][:, :, :, i] = r["backoff_x"].transpose(0, 2, 1)
            out["backoff_trajectory_u"][:, :, :, i] = r["backoff_u"].transpose(0, 2, 1)
            out["t_qp"][i], out["t_riccati"][i], out["t_jac"][i] = r["t_qp_ms"], r["t_riccati_ms"], r["t_jac_ms"]
            out["primal_infeasibility"][:, i] = r["primal_infeasibility"]
            out["success"][:, i] = r["success"]
            out["scp_iterations"][:, i] = r["scp_iterations"]
        return out

    def save_npz(self, path, out, b=0):
        """Write instance b with exactly the key set the reference's plot()/loaders read (main_rocket...:189-206, 218-241)."""
        m = self.m
        steps = out["state_trajectory"].shape[2]
        np.savez(path, state_trajectory=out["state_trajectory"][b], input_trajectory=out["input_trajectory"][b],
                 nominal_trajectory_x=out["nominal_trajectory_x"][b], nominal_trajectory_u=out["nominal_trajectory_u"][b],
                 backoff_trajectory_x=out["backoff_trajectory_x"][b], backoff_trajectory_u=out["backoff_trajectory_u"][b],
                 dt=0.05, g=m.g, nx=m.nx, nu=m.nu, simulation_time_steps=steps, N=self.N,
                 t_jac=out["t_jac"], t_qp=out["t_qp"], t_riccati=out["t_riccati"])
