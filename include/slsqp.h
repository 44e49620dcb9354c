/*
 * slsqp.h -- C ABI of the MI355X-native batched fast-SLS QP path (libslsqp_hip.so).
 *
 * Every entry point replaces a call the reference makes on its hot path (citations relative to the
 * reference repository antoineleeman/robust-nonlinear-mpc):
 *
 *   slsqp_create / destroy          fast_SLS.__init__                 solver/fast_SLS_jit.py:202-241
 *                                   (QP._assemble_structures + first OSQP setup, solver/qp_jit.py:77-192, 278-306)
 *   slsqp_set_costs                 OCP.__init__ weights              solver/ocp.py:8-38; SCP_SLS_jit.py:386-388
 *   slsqp_set_constraints           LTV(m,N) G,Gf,gf copies           dyn/LTV.py:17-32
 *   slsqp_update_dynamics           fast_SLS.update_dynamics_list     solver/fast_SLS_jit.py:250-273
 *                                   (QP.update_dynamics :518-576, offset_constraints :595-610)
 *   slsqp_update_linear_cost        fast_SLS.update_linear_cost       solver/fast_SLS_jit.py:575-576
 *   slsqp_solve                     fast_SLS.solve                    solver/fast_SLS_jit.py:278-312
 *   slsqp_get                       post_processing_solution          solver/fast_SLS_jit.py:602-646
 *   slsqp_reset                     reset_solver_to_zeros             solver/fast_SLS_jit.py:424-442
 *   slsqp_qp_update_data_mat/_vec,  module `osqp_generated`           solver/qp_jit.py:671-698 (push), :449-470 (solve)
 *   slsqp_qp_solve                  (update_data_mat / update_data_vec / solve)
 *   slsqp_sweep                     backward_solve + update_tightening kernels  solver/fast_SLS_jit.py:489-571
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; all arrays are C-contiguous fp64 (int32 for status/iter).
 *   - `loc` says where caller buffers live: SLSQP_HOST (library copies over PCIe) or SLSQP_DEVICE (device
 *     pointers on the handle's GPU, consumed/produced in place on the handle's stream).
 *   - the batch axis B (independent MPC instances) is always the leading axis.
 *   - return value: 0 ok, <0 API misuse / unsupported configuration (message via slsqp_last_error()).
 *     Per-instance outcomes never fail the call: see status[B] (SLSQP_ST_*).
 *   - one handle = one GPU + one HIP stream; distinct handles may be used from distinct threads.
 */
#ifndef SLSQP_H
#define SLSQP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct slsqp_handle slsqp_handle;

typedef struct {
    int nx, nu, nw; /* state / input / disturbance dims */
    int N;          /* horizon */
    int ni, ni_f;   /* stage / terminal inequality rows; must be 2(nx+nu) and 2nx (G=[I;-I], Gf=[I;-I]) */
} slsqp_dims;

enum { SLSQP_HOST = 0, SLSQP_DEVICE = 1 };

/* per-instance status */
enum {
    SLSQP_ST_SOLVED = 0,      /* KKT certificate met (polished active-set solution) */
    SLSQP_ST_SOLVED_IPM = 4,  /* interior-point tolerance met, polish rejected (solution accurate to opts.eps) */
    SLSQP_ST_MAX_ITER = 1,
    SLSQP_ST_INFEASIBLE = 2,  /* primal infeasible, seen before any work: the pinned x0 lies outside its own stage-0 box */
    SLSQP_ST_NUMERICAL = 3,
    SLSQP_ST_INFEASIBLE_CERT = 5 /* primal infeasible, certified by the interior point's multipliers (Farkas ray; the test OSQP applies for the
                                    reference, eps_prim_inf = 1e-4): E'nu + lambda_u - lambda_l ~ 0 with a negative support value */
};

typedef struct {
    int rti_steps;       /* >0: RTI mode, exactly that many fast-SLS steps then one final QP (fast_SLS_jit.py:280-296);
                            <=0: converge mode, up to max_sls_iter steps (:298-312) */
    int max_sls_iter;    /* MAX_ITER (30, fast_SLS_jit.py:206) */
    int qp_max_iter;     /* interior-point iteration cap per QP (default 60) */
    double qp_eps;       /* interior-point residual / complementarity tolerance before the polish (default 1e-6, relative to max(1,|q|inf));
                            if the polish then fails to certify, the interior point resumes down to 1e-9 and the polish is repeated */
    double conv_tol;     /* primal convergence test of check_convergence_socp (1e-3, fast_SLS_jit.py:594) */
    double eps_backoff;  /* epsilon_backoff (1e-10, fast_SLS_jit.py:205) */
    int want_K;          /* accepted and ignored: K (N,N+1,nu,nx) is always kept (the propagation of the sweep reads it back) */
    int warm_start;      /* 1 (default): the first QP of a call first tries an active-set polish from the instance's previous
                            certified solution (KKT-verified, falls back to the interior point); later QPs of a call always do */
    int warm_rounds;     /* active-set correction rounds a warm attempt may use before falling back (default 20) */
    int max_scp_iter;    /* MAX_ITER_SCP (100, SCP_SLS_jit.py:47): cap of the SCP loop of slsqp_cl_step in converge mode (rti <= 0) */
    double scp_eps;      /* epsilon_convergence (1e-10, SCP_SLS_jit.py:29): SCP converged when |delta_vec|inf < scp_eps */
    int precision;       /* 0 (default): fp64 throughout.  1: mixed -- block factorisations, stored inverses and substitutions in fp32,
                            right-hand sides / residuals / KKT certificate in fp64, two more refinement solves per polish; instances
                            that do not certify are solved again in fp64 (BASELINE config 3, "fp32 vs fp64") */
    int as_first;        /* 1: a cold QP solve first runs the active-set iteration from the empty set (round 0 = the equality-constrained
                            optimum), certificate-checked like every polish, and only falls back to the interior point when that fails; a failed warm
                            attempt is followed by that attempt from the empty set too.  2 (default): as 1, but a failed warm attempt goes straight to
                            the interior point (the retry from the empty set rescued 0.08 % of the failed warm attempts where it was measured).
                            0: no active-set attempt before the interior point */
    int as_rounds;       /* correction rounds such an attempt may use (default 24) */
    int as_max_viol;     /* an active-set attempt is abandoned when one of its solves leaves more violated bounds than this (default 64), or more
                            than twice the previous round's + 8: a set that pins both ends of a dynamics row makes the solve blow up */
    int ipm_restart;     /* 1 (default): QPs after the first of a fast-SLS call (same A, B, q, tightened bounds) whose warm active-set attempt
                            fails restart the interior point from a copy of the first QP's iterate at mu ~ 1e-3 |q|inf instead of from scratch */
    int time_kernels;    /* 0 (default): no per-launch timing.  1: HIP event pairs around every launch of the dominant QP kernel on the
                            handle's stream, read with slsqp_kernel_timing (bench.py's roofline leg) */
    int as_warm_max_set; /* 28 (default): the last QP of a call does not start its warm active-set attempt from the first QP's set when that set has more
                            active bounds than this (0 = no limit): it starts from the previous call's last set (as_warm_last) or, without one, the interior point */
    int as_warm_last;    /* 1 (default): where as_warm_max_set rules out the first QP's set (or the first QP left no certified set), the last QP of a call starts
                            from the certified active set of the PREVIOUS call's last QP, moved one stage with the horizon after slsqp_cl_step's shift;
                            2: whenever such a set exists; 0: never */
    int fuse_rti;        /* 1 (default): an RTI solve with rti_steps = 1 in fp64 (the rocket script's setting) of a batch with B (N+1) >= 3072 runs as ONE launch
                            in which every wavefront takes its instance through QP -> eta -> Riccati / propagation -> tightened bounds -> QP (k_rti_chain): no
                            batch-wide barrier behind a QP, so a slow instance only delays itself.  Smaller batches keep the separate launches (their SLS
                            propagation runs N+1 columns of an instance side by side: lower latency).  2: the chain whatever the batch; 0: never.
                            Same arithmetic, same results bit for bit. */
    int cl_persistent;   /* 1 (default): slsqp_cl_run is ONE persistent launch (k_cl_loop): wavefronts take instances from a device-side FIFO and run one
                            whole MPC step each time (shift, linearisation, RTI chain, nominal update, plant), so no wave slot ever waits for a round or
                            for another instance; budget_ms / cut_frac are ignored and *rounds_out = 1.  0: the round-based loop described at
                            slsqp_cl_run.  Same results bit for bit either way. */
} slsqp_opts;

void slsqp_default_opts(slsqp_opts *o);
const char *slsqp_last_error(void);
const char *slsqp_version(void);

slsqp_handle *slsqp_create(const slsqp_dims *d, int batch, int device);
void slsqp_destroy(slsqp_handle *h);

/* Q (nx,nx) R (nu,nu) Qf (nx,nx): must be diagonal (all reference scripts use np.diag); Q_reg/R_reg/Q_reg_f likewise.
   Host pointers. Batch-constant. */
int slsqp_set_costs(slsqp_handle *h, const double *Q, const double *R, const double *Qf, const double *Q_reg,
                    const double *R_reg, const double *Q_reg_f);
/* G (ni,nx+nu), Gf (ni_f,nx), gf (ni_f): host pointers.  G = [I;-I], Gf = [I;-I] (every plant of the reference) enables all entry points;
   any other G / Gf is accepted for the sweep-level boundary (slsqp_sweep) only -- the QP-level entry points then refuse.  gf is the RAW terminal bound that the
   tightened QP uses (fast_SLS_jit.py:524,568). */
int slsqp_set_constraints(slsqp_handle *h, const double *G, const double *Gf, const double *gf);

/* A (B,N,nx,nx)  Bm (B,N,nx,nu)  E (N+1,nx,nw) shared by the batch (NULL keeps the previous one)
   g (B,N,ni) shifted stage bounds g_k = g - G [z_k;v_k];  g_N (B,ni_f) shifted terminal bound gf - Gf z_N
   c (B,N,nx) dynamics offsets c_k = f(z_k,v_k) - z_{k+1}.
   Effect = QP.update_dynamics + offset_constraints: ubg <- [-c_k + eps ; g_k + eps]..., g_N + eps ; lbg <- [-c_k - eps ; -inf]. */
int slsqp_update_dynamics(slsqp_handle *h, const double *A, const double *Bm, const double *E, const double *g,
                          const double *g_N, const double *c, int loc);
int slsqp_update_linear_cost(slsqp_handle *h, const double *q, int loc); /* q (B,n) */

/* x0 (B,nx): the argument of fast_SLS.solve, i.e. x_nom0 - x_meas.  Runs the whole fast-SLS iteration on the GPU. */
int slsqp_solve(slsqp_handle *h, const double *x0, int loc, const slsqp_opts *opts);

/* Fetch a result array by name into `out` (host or device).  Names and shapes (B leading):
   primal_vec (n) dual_vec (m-nx) cost_nominal (1) cost_tube (1; SLS.eval_cost of the last sweep, util/SLS.py:38-46) status[int32] (1) qp_iters[int32] (1) iteration_number[int32] (1)
   beta (N,N,ni) beta_f (N+1,ni_f) backoff (N,ni) backoff_f (ni_f) backoff_x (N+1,nx) backoff_u (N,nu)
   eta (N,N,ni) eta_f (N+1,ni_f) K (N,N+1,nu,nx) ubg (m-nx) lbg (m-nx) kkt (8) pin_dual (nx) success[int32] (1)
   and the current problem data: A (N,nx,nx) Bm (N,nx,nu) c (N,nx) g (N,ni) gN (ni_f) q (n) */
int slsqp_get(slsqp_handle *h, const char *name, void *out, int loc);
/* bytes slsqp_get copies per instance for `name` (the caller's buffer must hold batch x that), or -1 for an unknown name */
long long slsqp_result_bytes(slsqp_handle *h, const char *name);
/* Overwrite a named array: ubg, lbg (QP.update_ubg / reset_lbg, qp_jit.py:578-593, poked by SCP_SLS_jit.py:86-99), q, nominal_x, nominal_u,
   x_meas.  Same shapes as slsqp_get. */
int slsqp_set(slsqp_handle *h, const char *name, const void *src, int loc);
int slsqp_reset(slsqp_handle *h);
int slsqp_sync(slsqp_handle *h);

/* ---- the step in front of the path: batched linearisation (SCP_SLS.update_jacobian, solver/SCP_SLS_jit.py:251-366) ---------
   model_id: 0 pendulum, 1 quadrotor, 2 rocket (ODEs of dyn/{pendulum,quadrotor,rocket}.py, RK4 h=0.05, dyn/model.py:15-34); g_raw (ni): the plant's stage bound g.
   slsqp_linearize: X (B,N+1,nx), U (B,N,nu) nominal trajectories (stage-major) -> A,B (forward-mode AD through RK4),
   c_k = f(x_k,u_k) - x_{k+1}, g_k = g - G[x_k;u_k], g_N = gf - Gf x_N, q = 2 H y_nom, then the un-tightened bounds; E is untouched
   (set it once with slsqp_update_dynamics or slsqp_set_E). Equivalent to update_dynamics + update_linear_cost. */
int slsqp_set_model(slsqp_handle *h, int model_id, const double *g_raw);
int slsqp_set_E(slsqp_handle *h, const double *E, int loc);   /* E (N+1,nx,nw) */
int slsqp_linearize(slsqp_handle *h, const double *X, const double *U, int loc);

/* ---- the caller after the path: SCP update, warm-start shift, plant step (SCP_SLS.socp_step solver/SCP_SLS_jit.py:404-473,
   reset_warm_start :500-551, expe/main_rocket_robust_closed_loop.py:149-182) -- a whole closed-loop MPC step with no host round trip.
   slsqp_cl_init: x_meas (B,nx); X_nom (B,N+1,nx), U_nom (B,N,nu) initial nominal, or NULL,NULL -> roll-out of the plant from x_meas
   under the constant input u_init (nu) (the reference's IPOPT initialiser is out of scope; pass its result here to reproduce it).
   slsqp_cl_step: rti > 0: exactly rti SCP iterations (scripts: rocket 1, pendulum/quadrotor 3); rti <= 0: SCP_SLS's default
   converge mode (rti = -1, SCP_SLS_jit.py:20-21,113-135): iterate every instance until |delta_vec|inf < opts.scp_eps, at most
   opts.max_scp_iter times; an instance whose fast-SLS step fails leaves the loop (:118-119).  w (B,nx) disturbance sample or NULL.
   Results via slsqp_get: nominal_x (N+1,nx) nominal_u (N,nu) x_meas (nx) u0 (nu) scp_success[int32] (1) scp_iterations[int32] (1)
   scp_delta_max (1) primal_infeasibility (1; max_k,i (ddyn(x_k,u_k) - x_{k+1})_i of the updated nominal, SCP_SLS_jit.py:449-456) + all names of the fast-SLS result (for each instance: of its last fast-SLS solve). */
int slsqp_cl_init(slsqp_handle *h, const double *x_meas, const double *X_nom, const double *U_nom, const double *u_init, int loc);
int slsqp_cl_step(slsqp_handle *h, int rti, const double *w, int loc, const slsqp_opts *opts);
/* The whole closed loop with the instances advancing INDEPENDENTLY (rti = 1, one fast-SLS step, fp64, fuse_rti: the rocket script's setting):
   `steps` MPC steps of every instance, W (steps,B,nx) disturbance samples or NULL.  Per instance the same operations in the same order as `steps`
   calls of slsqp_cl_step (identical results), but in rounds: in a round an instance begins its next MPC step or resumes the QP solve that the
   previous round's deadline suspended; a chain still running budget_ms after its launch started (budget_ms <= 0: no time limit), or still running when
   cut_frac of the round's participants have finished (0 < cut_frac < 1; else off), suspends itself and continues in the next round.
   No instance waits for the slowest one of its step.  Call after slsqp_cl_init (+ slsqp_nominal_solve); per-step results through the device-side
   log (slsqp_cl_log with max_steps >= steps, before slsqp_cl_init) and `log_qp_stats`[int32] (steps,2,8).  *rounds_out (may be NULL): rounds taken.
   With opts.cl_persistent (the default) there are no rounds at all: see slsqp_opts. */
int slsqp_cl_run(slsqp_handle *h, int steps, const double *W, int loc, const slsqp_opts *opts, double budget_ms, double cut_frac, int *rounds_out);
/* wave statistics of the last persistent slsqp_cl_run: [0] wavefronts launched, [1] sum over the MPC steps of the time a wavefront spent on them (ms),
   [2] MPC steps run, [3] duration of the launch (ms, HIP events; 0 without opts.time_kernels).  [1] / ([0] x [3]) = how busy the queue kept the waves. */
#define SLSQP_CL_RUN_STATS_LEN 4
int slsqp_cl_run_stats(slsqp_handle *h, double *out, int len);
/* Device-side log of the closed loop: every following slsqp_cl_step stores what the scripts keep per MPC step
   (expe/main_rocket_robust_closed_loop.py:160-178) in entry `step` of (B, max_steps, ...) device buffers, so a Monte-Carlo run makes no
   host round trip per step.  slsqp_get names (per instance): log_state (S,nx) log_u0 (S,nu) log_nominal_x (S,N+1,nx) log_nominal_u (S,N,nu)
   log_backoff_x (S,N+1,nx) log_backoff_u (S,N,nu) log_success[int32] (S) log_scp_iterations[int32] (S) log_primal_infeasibility (S).
   slsqp_cl_init restarts at entry 0. */
int slsqp_cl_log(slsqp_handle *h, int max_steps);

/* ---- nominal-trajectory initialiser: replaces the reference's IPOPT call for the first MPC step
   (SCP_SLS.solve_nominal_trajectory solver/SCP_SLS_jit.py:161-188, NLP of solver/nlp.py:158-217:
   min sum x'Qx + u'Ru + xN'Qf xN  s.t. x+ = ddyn(x,u), G[x;u] <= g, Gf xN <= gf, x0 = x_meas).
   Improves the nominal held by the handle (slsqp_cl_init: the caller's guess or the roll-out) for every instance by trust-region
   sequential convex programming on the path's own QP kernel, until the step is below tol (default 1e-7) with feasible dynamics and
   box, at most max_qp QP solves (default 120); rho (default 1e3) weighs constraint violation in the merit function.
   Results via slsqp_get: nominal_x, nominal_u, nlp_status[int32] (0 converged to a KKT point, 1 max_qp reached, 2 failed:
   no acceptable step / QP infeasible), nlp_iterations[int32] (accepted steps), nlp_info (12): w, kappa, kappa0, cost, defect l1, box
   violation l1, last ratio, last |step|inf.  IPOPT output is not available offline: parity for this entry point is "unpinned";
   tests certify the NLP's KKT conditions independently. */
int slsqp_nominal_solve(slsqp_handle *h, int max_qp, double tol, double rho, const slsqp_opts *opts);

/* ---- QP-level boundary: mirrors the three calls of the reference's generated module ----------------------- */
/* P_x (B,nnzP): CSC data of triu(2P) (diagonal => nnzP = n);  A_x (B,nnzA): CSC data of the (m x n) constraint
   matrix INCLUDING the x0-pin rows, sorted indices, pattern of qp_jit.py:77-192 with G=[I;-I]. */
int slsqp_qp_nnz(const slsqp_dims *d, int *n, int *m, int *nnzP, int *nnzA);
int slsqp_qp_update_data_mat(slsqp_handle *h, const double *P_x, const double *A_x, int loc);
int slsqp_qp_update_data_vec(slsqp_handle *h, const double *q, const double *l, const double *u, int loc); /* (B,n) (B,m) (B,m) */
/* x (B,n), y (B,m) OSQP sign convention, status (B), iters (B) */
int slsqp_qp_solve(slsqp_handle *h, double *x, double *y, int *status, int *iters, int loc, const slsqp_opts *opts);

/* ---- sweep-level boundary ---------------------------------------------------------------------------------- */
/* eta (B,N,N,ni) eta_f (B,N+1,ni_f) in; K/beta/beta_f/backoff/backoff_f out (any may be NULL). Uses the handle's A,B,E. */
int slsqp_sweep(slsqp_handle *h, const double *eta, const double *eta_f, double *K, double *beta, double *beta_f,
                double *backoff, double *backoff_f, int loc);

/* Both timing queries take the LENGTH of the caller's buffer and refuse (return < 0, nothing written) a buffer shorter than what they
   write: SLSQP_TIMING_LEN / SLSQP_KERNEL_TIMING_LEN doubles.  (Round 2 took bare pointers; a caller that still sized its buffer for four
   values when a fifth was added corrupted its own heap.)  Longer buffers are fine: only the first SLSQP_*_LEN entries are written. */
#define SLSQP_TIMING_LEN 5
#define SLSQP_KERNEL_TIMING_LEN 8
/* elapsed GPU time (ms) of the kernels launched by the last slsqp_solve / slsqp_qp_solve / slsqp_sweep / slsqp_cl_step call,
   measured with HIP events on the handle's stream: [0] total, [1] QP kernel (k_qp_solve), [2] sweep kernels, [3] other,
   [4] linearisation of the last slsqp_cl_step (the reference's t_jac, SCP_SLS_jit.py:268,339-341). */
int slsqp_last_timing(slsqp_handle *h, double *ms, int len);
/* accumulated since the last call (opts.time_kernels = 1 for [0], [1]): [0] total ms of the launches of the dominant QP kernel (k_qp_solve: one
   launch per QP solve) from HIP events around each launch on the handle's stream, [1] number of those launches, [2] instances re-solved in fp64
   after a mixed-precision attempt, and device counters of the work done: [3] instance forward sweeps, [4] how many of them factorised,
   [5] stages factorised (a factorising sweep of an active-set round re-does only the stages from the first changed one on), [6] QP solves that
   ran at least one block solve, [7] block solves that ended after their forward sweep (residual check of an already certified solve: no
   backward sweep); resets the accumulators. */
int slsqp_kernel_timing(slsqp_handle *h, double *out, int len);
void *slsqp_stream(slsqp_handle *h); /* hipStream_t, for callers that share device buffers */
/* Diagnostic: one wavefront runs one of the wave-level building blocks of the kernels (csrc/wave_la.hpp: the MFMA block products, the fused
   product pair of the SLS propagation, the Gauss-Jordan SPD inverse, the D_k assembly) on packed row-major host operands; (nx,nu) = (17,4)
   or (13,4).  Cases and operand order: slsqp_api.hip, k_selftest.  Used by tests/test_gpu_parity.py::test_wave_level_building_blocks. */
int slsqp_selftest(int nx, int nu, int which, const double *in, int n_in, double *out, int n_out);

#ifdef __cplusplus
}
#endif
#endif
