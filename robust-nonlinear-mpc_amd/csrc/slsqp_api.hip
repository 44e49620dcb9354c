// slsqp_api.hip -- C-ABI host side of libslsqp_hip.so (see include/slsqp.h for the contract and the reference
// call sites each entry point replaces).  The fast-SLS control flow of fast_SLS.solve
// (solver/fast_SLS_jit.py:278-327) is run here as a short sequence of batched kernel launches with
// per-instance masks, so one infeasible or converged instance never stalls or fails the batch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../include/slsqp.h"
#include "slsqp_kernels.hpp"

using namespace slsqp;

static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return -1; }
#define HIPCHK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_));         \
    } while (0)

extern "C" const char *slsqp_last_error(void) { return g_err.c_str(); }
extern "C" const char *slsqp_version(void) { return "slsqp-hip 0.1 (gfx950)"; }

extern "C" void slsqp_default_opts(slsqp_opts *o) {
    o->rti_steps = 1; o->max_sls_iter = 30; o->qp_max_iter = 60; o->qp_eps = 1e-6; o->conv_tol = 1e-3;
    o->eps_backoff = 1e-10; o->want_K = 1; o->warm_start = 1; o->warm_rounds = 20;
    o->max_scp_iter = 100; o->scp_eps = 1e-10; o->precision = 0; o->time_kernels = 0; o->as_first = 2; o->as_rounds = 24; o->as_max_viol = 64; o->ipm_restart = 1; o->as_warm_max_set = 28; o->as_warm_last = 1; o->fuse_rti = 1; o->cl_persistent = 1;
}

struct slsqp_handle {
    slsqp_dims d;
    int B, dev, n, m, mb, nz;
    hipStream_t st;
    // problem data
    double *A, *Bm, *E, *g, *gN, *c, *q, *x0val, *gf_raw, *g_raw, *cst;
    int model_id;
    double *Xn, *Un, *xmeas, *x0arg, *u0, *wbuf, *u_init; int cl_steps;  // cst: Qd Rd Qfd Qregd Rregd Qregfd packed
    double *ubg, *lbg;
    // results / state
    double *primal, *dual, *cost, *pin_dual, *kkt, *prev_primal, *Linv, *ws, *qpstate;
    double *eta, *eta_f, *beta, *beta_f, *backoff, *backoff_f, *backoff_x, *backoff_u, *K;
    int *status, *iters, *itnum, *has_prev, *conv, *alive, *mask, *success, *infeas, *counter;
    int *scp_active, *scp_success, *scp_iters, *pending_reset, *scp_upd; double *scp_dmax;
    int horizon_shifted = 0;
    // timeline: event pairs recorded on the stream with a role (0 QP, 1 sweep, 2 whole solve, 3 linearisation, 4 fused RTI chain); read after a synchronisation
    std::vector<hipEvent_t> tl; std::vector<int> tl_role; int tl_n = 0; double tl_acc[5] = {0, 0, 0, 0, 0};     // role 4: fused RTI chain launches (k_rti_chain)
    // slsqp_cl_run: per-instance progress through the closed loop (step counter, suspended-solve state, round masks), per-instance call ids, the launch's
    // start-time word, the per-step copy of qp_stats
    int *cl_stepno = nullptr, *cl_lag = nullptr, *cl_begin = nullptr, *cl_runm = nullptr, *cl_done = nullptr, *cl_skipb = nullptr, *cl_skip_begin = nullptr, *qplog = nullptr;
    double *call_ids = nullptr, *cl_W = nullptr; size_t cl_W_doubles = 0; unsigned long long *t0word = nullptr; int qplog_steps = 0;
    int *clq_slots = nullptr, *clq_ctl = nullptr; unsigned clq_cap = 0; int cl_loop_waves = 0; double cl_loop_ms = 0.0; unsigned long long *cl_busy = nullptr, *cl_tbegin = nullptr, cl_busy_host[16] = {};      // k_cl_loop: instance FIFO (slots; head, tail, avail, err), wave life-time counters
    bool cl_round = false; unsigned long long cl_budget = 0; int cl_total_steps = 0; unsigned cl_cut_count = 0xFFFFFFFFu;
    unsigned long long *chain_times = nullptr, *chain_times_host = nullptr;   // (B,4) in-kernel wall-clock ticks per instance; pinned copy of instance 0's
    int *qp_diag = nullptr;     // QP_DIAG_SPAN builds only
    double *nom_st; int *nom_need_lin, *nom_status, *nom_iters;
    int *retry; int mx_retry, mx_retry_total;
    double *Kc, *Aclc;          // (B,N,nu,nx), (B,N,nx,nx): K_k and A_k + B_k K_k of the shared Riccati recursion (k_sweep_ric1 -> k_sweep_prop)
    double *lin_stage, *lin_tape;   // (B,N,3,nx) intermediate RK4 stage points of the linearisation (k_lin_val -> k_lin_tan); (B,N,4,NT_MAX) its transcendental values
    double call_id;             // counts fast-SLS calls (validity of the interior-point iterate copies, QpArgs::call_id)
    int *qpstat;                // (B,2,8) per-QP statistics, see QpArgs::qpstat
    int *stale;                 // (B) bit 0: eta / eta_f, bit 1: K hold values from before the last slsqp_reset (zeroed lazily on slsqp_get)
    double *pinf; double t_jac;  // primal_infeasibility of the last SCP update (SCP_SLS_jit.py:449-456); linearisation time of the last cl_step
    // qp-level CSC maps
    int *mapA, *mapB;  // CSC offsets of A_k[i][j] / B_k[i][j]
    double *stage;     // staging buffer for host<->device transfers
    size_t stage_bytes;
    bool have_costs, have_cons, have_dyn;
    bool beta_inited;           // beta / beta_f have been filled with eps once (k_init_backoff); later solves only repair swept instances (last part of k_after_qp)
    bool general_G;             // G, Gf are not [I;-I]: only the sweep-level boundary (slsqp_sweep) is available
    double *Gd, *Gfd;           // device copies of G (ni, nx+nu) and Gf (ni_f, nx) when general_G
    hipEvent_t ev[10];
    std::vector<hipEvent_t> kev;   // event pairs around every k_qp_solve launch since the last harvest (only with opts.time_kernels)
    int n_kev; bool time_kernels;
    double t_total, t_qp, t_sweep, t_fwd; int n_fwd; double fwd_inst;   // fwd_inst: sum over launches of instances that did work
    std::map<std::string, std::pair<void *, size_t>> named;  // name -> (device ptr, bytes per instance)
    unsigned long long *inst_launches;                       // device counters of k_qp_solve's work (QpArgs::inst_launches; roofline accounting)
    double *ct_part, *cost_tube;                             // sweep's per-column parts of cost_tube^2 and the result
    std::vector<void *> owned, log_owned;                    // device buffers of the handle / of its closed-loop log
    int log_steps;                                           // device-side closed-loop log (slsqp_cl_log): capacity in MPC steps, 0 = off
    double *lg_x, *lg_u, *lg_bx, *lg_bu, *lg_state, *lg_u0, *lg_pinf; int *lg_succ, *lg_it;
};

static Costs costs_of(slsqp_handle *h) {
    const int nx = h->d.nx, nu = h->d.nu;
    Costs c;
    c.Qd = h->cst; c.Rd = c.Qd + nx; c.Qfd = c.Rd + nu; c.Qregd = c.Qfd + nx; c.Rregd = c.Qregd + nx; c.Qregfd = c.Rregd + nu; c.prox = 0.0;
    return c;
}

template <typename T>
static int dalloc(std::vector<void *> &owned, T **p, size_t count) {
    *p = nullptr;
    HIPCHK(hipMalloc((void **)p, count * sizeof(T) + 64));
    owned.push_back((void *)*p);
    HIPCHK(hipMemset(*p, 0, count * sizeof(T)));
    return 0;
}
static void free_all(std::vector<void *> &owned) { for (void *p : owned) if (p) hipFree(p); owned.clear(); }

// (nx, nu) pairs the QP / sweep kernels are instantiated for: the reference's three plants, plus whatever the build adds with
//   -DSLSQP_EXTRA_DIMS="X(6,2) X(9,3)"      (limits of the single-wave kernels: nx <= 17, nu <= 4, nx + nu <= 21)
// The plant-specific entry points (slsqp_set_model / slsqp_linearize / slsqp_cl_*) exist for the three plants only.
#ifndef SLSQP_EXTRA_DIMS
#define SLSQP_EXTRA_DIMS
#endif
#define SLSQP_DIM_LIST X(4, 1) X(13, 4) X(17, 4) SLSQP_EXTRA_DIMS
static bool supported_dims(int nx, int nu) {
#define X(NX_, NU_) if (nx == NX_ && nu == NU_) return true;
    SLSQP_DIM_LIST
#undef X
    return false;
}
static std::string dims_list() {
    std::string r;
#define X(NX_, NU_) r += " (" #NX_ "," #NU_ ")";
    SLSQP_DIM_LIST
#undef X
    return r;
}

extern "C" int slsqp_qp_nnz(const slsqp_dims *d, int *n, int *m, int *nnzP, int *nnzA) {
    const int nz = d->nx + d->nu;
    *n = nz * d->N + d->nx;
    *m = d->N * (d->nx + d->ni) + d->ni_f + d->nx;
    *nnzP = *n;
    *nnzA = d->N * (d->nx * (nz + 1) + 2 * nz) + 3 * d->nx;
    return 0;
}

extern "C" slsqp_handle *slsqp_create(const slsqp_dims *d, int batch, int device) {
    if (!supported_dims(d->nx, d->nu)) { fail("unsupported (nx,nu): this build instantiates the kernels for" + dims_list() + " (add pairs with -DSLSQP_EXTRA_DIMS)"); return nullptr; }
    if (d->ni < 1 || d->ni_f < 1 || d->ni > 4096 || d->ni_f > 4096) { fail("need 1 <= ni, ni_f <= 4096"); return nullptr; }
    if (d->nw < 1 || d->nw > d->nx) { fail("need 1 <= nw <= nx (E is kept zero-padded to nx columns: the padding adds nothing to any row norm of Phi)"); return nullptr; }
    if (d->N < 1 || d->N > 64 || batch < 1) { fail("need 1 <= N <= 64 and batch >= 1"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fail("no HIP device visible: libslsqp_hip has no CPU fallback"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail("hipSetDevice failed"); return nullptr; }
    slsqp_handle *h = new slsqp_handle();
    h->d = *d; h->B = batch; h->dev = device;
    const int nx = d->nx, nu = d->nu, N = d->N, ni = d->ni, nif = d->ni_f, nw = d->nw;
    h->nz = nx + nu; h->n = h->nz * N + nx; h->mb = N * (nx + ni) + nif; h->m = h->mb + nx;
    const size_t B = batch;
    if (hipStreamCreate(&h->st) != hipSuccess) { fail("hipStreamCreate"); delete h; return nullptr; }
    int rc = 0;
    rc |= dalloc(h->owned, &h->A, B * N * nx * nx); rc |= dalloc(h->owned, &h->Bm, B * N * nx * nu); rc |= dalloc(h->owned, &h->E, (size_t)(N + 1) * nx * nx);     // (N+1, nx, nx): E (N+1, nx, nw) zero-padded to nx columns
    (void)nw;
    rc |= dalloc(h->owned, &h->g, B * N * ni); rc |= dalloc(h->owned, &h->gN, B * nif); rc |= dalloc(h->owned, &h->c, B * N * nx); rc |= dalloc(h->owned, &h->q, B * h->n);
    rc |= dalloc(h->owned, &h->x0val, B * nx); rc |= dalloc(h->owned, &h->gf_raw, (size_t)nif); rc |= dalloc(h->owned, &h->g_raw, (size_t)ni);
    rc |= dalloc(h->owned, &h->Xn, B * (N + 1) * nx); rc |= dalloc(h->owned, &h->Un, B * N * nu); rc |= dalloc(h->owned, &h->xmeas, B * nx); rc |= dalloc(h->owned, &h->x0arg, B * nx);
    rc |= dalloc(h->owned, &h->u0, B * nu); rc |= dalloc(h->owned, &h->wbuf, B * nx); rc |= dalloc(h->owned, &h->u_init, (size_t)nu); h->cl_steps = 0; rc |= dalloc(h->owned, &h->cst, (size_t)(3 * nx + 2 * nu) * 2);
    rc |= dalloc(h->owned, &h->ubg, B * h->mb); rc |= dalloc(h->owned, &h->lbg, B * h->mb);
    rc |= dalloc(h->owned, &h->primal, B * h->n); rc |= dalloc(h->owned, &h->dual, B * h->mb); rc |= dalloc(h->owned, &h->cost, B); rc |= dalloc(h->owned, &h->pin_dual, B * nx);
    rc |= dalloc(h->owned, &h->kkt, B * 8); rc |= dalloc(h->owned, &h->prev_primal, B * h->n); rc |= dalloc(h->owned, &h->Linv, B * N * nx * nx); rc |= dalloc(h->owned, &h->ws, B * qp_ws_doubles(h->n, N, nx)); rc |= dalloc(h->owned, &h->qpstate, B * 40);
    rc |= dalloc(h->owned, &h->eta, B * N * N * ni); rc |= dalloc(h->owned, &h->eta_f, B * (N + 1) * nif); rc |= dalloc(h->owned, &h->beta, B * N * N * ni);
    rc |= dalloc(h->owned, &h->beta_f, B * (N + 1) * nif); rc |= dalloc(h->owned, &h->backoff, B * N * ni); rc |= dalloc(h->owned, &h->backoff_f, B * nif);
    rc |= dalloc(h->owned, &h->backoff_x, B * (N + 1) * nx); rc |= dalloc(h->owned, &h->backoff_u, B * N * nu); rc |= dalloc(h->owned, &h->K, B * N * (N + 1) * nu * nx);
    rc |= dalloc(h->owned, &h->status, B); rc |= dalloc(h->owned, &h->iters, B); rc |= dalloc(h->owned, &h->itnum, B); rc |= dalloc(h->owned, &h->has_prev, B); rc |= dalloc(h->owned, &h->conv, B);
    rc |= dalloc(h->owned, &h->alive, B); rc |= dalloc(h->owned, &h->mask, B); rc |= dalloc(h->owned, &h->success, B); rc |= dalloc(h->owned, &h->infeas, B); rc |= dalloc(h->owned, &h->counter, (size_t)4);
    rc |= dalloc(h->owned, &h->scp_active, B); rc |= dalloc(h->owned, &h->scp_success, B); rc |= dalloc(h->owned, &h->scp_iters, B); rc |= dalloc(h->owned, &h->pending_reset, B); rc |= dalloc(h->owned, &h->scp_dmax, B);
    rc |= dalloc(h->owned, &h->retry, B); h->mx_retry = 0; h->mx_retry_total = 0;
    rc |= dalloc(h->owned, &h->nom_st, B * 12); rc |= dalloc(h->owned, &h->nom_need_lin, B); rc |= dalloc(h->owned, &h->nom_status, B); rc |= dalloc(h->owned, &h->nom_iters, B);
    rc |= dalloc(h->owned, &h->mapA, (size_t)N * nx * nx); rc |= dalloc(h->owned, &h->mapB, (size_t)N * nx * nu);
    rc |= dalloc(h->owned, &h->inst_launches, (size_t)8); rc |= dalloc(h->owned, &h->chain_times, B * 4);
    rc |= dalloc(h->owned, &h->cl_stepno, B); rc |= dalloc(h->owned, &h->cl_lag, B); rc |= dalloc(h->owned, &h->cl_begin, B); rc |= dalloc(h->owned, &h->cl_runm, B); rc |= dalloc(h->owned, &h->cl_done, B);
    { unsigned cap = 1; while (cap < 4u * (unsigned)B) cap <<= 1; h->clq_cap = cap; rc |= dalloc(h->owned, &h->clq_slots, (size_t)cap); rc |= dalloc(h->owned, &h->clq_ctl, (size_t)8); rc |= dalloc(h->owned, &h->cl_busy, (size_t)16); rc |= dalloc(h->owned, &h->cl_tbegin, (size_t)B); }
    rc |= dalloc(h->owned, &h->cl_skipb, B); rc |= dalloc(h->owned, &h->call_ids, B); rc |= dalloc(h->owned, &h->t0word, (size_t)2); rc |= dalloc(h->owned, &h->ct_part, B * (N + 1)); rc |= dalloc(h->owned, &h->cost_tube, B);
    rc |= dalloc(h->owned, &h->lin_stage, B * N * 3 * nx); rc |= dalloc(h->owned, &h->lin_tape, B * N * 4 * dyn::NT_MAX); rc |= dalloc(h->owned, &h->Kc, B * N * nu * nx); rc |= dalloc(h->owned, &h->Aclc, B * N * nx * nx); rc |= dalloc(h->owned, &h->qpstat, B * 16); rc |= dalloc(h->owned, &h->Gd, (size_t)ni * (nx + nu)); rc |= dalloc(h->owned, &h->Gfd, (size_t)nif * nx); h->general_G = false; h->beta_inited = false; rc |= dalloc(h->owned, &h->stale, B); rc |= dalloc(h->owned, &h->pinf, B); rc |= dalloc(h->owned, &h->scp_upd, B); h->t_jac = 0;
    h->log_steps = 0; h->lg_x = h->lg_u = h->lg_bx = h->lg_bu = h->lg_state = h->lg_u0 = h->lg_pinf = nullptr; h->lg_succ = h->lg_it = nullptr;
    if (rc) { free_all(h->owned); hipStreamDestroy(h->st); delete h; return nullptr; }
    if (hipHostMalloc((void **)&h->chain_times_host, 4 * sizeof(unsigned long long)) != hipSuccess) h->chain_times_host = nullptr;
    else memset(h->chain_times_host, 0, 4 * sizeof(unsigned long long));
    for (auto &e : h->ev) hipEventCreate(&e);
    h->tl.resize(2 * 256); for (auto &e : h->tl) hipEventCreate(&e);
    h->tl_role.assign(256, 0); h->tl_n = 0;
    h->kev.resize(2 * 256); for (auto &e : h->kev) hipEventCreate(&e);
    h->n_kev = 0; h->t_fwd = 0; h->n_fwd = 0; h->fwd_inst = 0; h->time_kernels = false; h->call_id = 0;
    // CSC offsets of the reference's frozen pattern (qp_jit.py:101-123,178-186; columns sorted by row)
    {
        std::vector<int> mA((size_t)N * nx * nx), mB((size_t)N * nx * nu);
        int p = 0;
        for (int k = 0; k <= N; k++) {
            for (int j = 0; j < nx; j++) {      // column of x_k[j]
                if (k >= 1) p += 1;              // -I entry in dynamics row block k-1
                if (k < N) { for (int i = 0; i < nx; i++) mA[((size_t)k * nx + i) * nx + j] = p + i; p += nx; p += 2; }
                else p += 2;                     // terminal +1 / -1
                if (k == 0) p += 1;              // x0 pin row (last rows of the matrix)
            }
            if (k < N)
                for (int a = 0; a < nu; a++) {   // column of u_k[a]
                    for (int i = 0; i < nx; i++) mB[((size_t)k * nx + i) * nu + a] = p + i;
                    p += nx; p += 2;
                }
        }
        hipMemcpyAsync(h->mapA, mA.data(), mA.size() * sizeof(int), hipMemcpyHostToDevice, h->st);
        hipMemcpyAsync(h->mapB, mB.data(), mB.size() * sizeof(int), hipMemcpyHostToDevice, h->st);
        hipStreamSynchronize(h->st);
    }
    h->stage = nullptr; h->stage_bytes = 0;
    h->have_costs = h->have_cons = h->have_dyn = false; h->model_id = -1;
    h->t_total = h->t_qp = h->t_sweep = 0;
    auto reg = [&](const char *nm, void *p, size_t bytes) { h->named[nm] = {p, bytes}; };
    reg("primal_vec", h->primal, sizeof(double) * h->n); reg("dual_vec", h->dual, sizeof(double) * h->mb); reg("cost_nominal", h->cost, sizeof(double));
    reg("status", h->status, sizeof(int)); reg("qp_iters", h->iters, sizeof(int)); reg("iteration_number", h->itnum, sizeof(int));
    reg("success", h->success, sizeof(int)); reg("scp_success", h->scp_success, sizeof(int)); reg("scp_iterations", h->scp_iters, sizeof(int));
    reg("scp_delta_max", h->scp_dmax, sizeof(double));
    reg("nlp_status", h->nom_status, sizeof(int)); reg("nlp_iterations", h->nom_iters, sizeof(int)); reg("nlp_info", h->nom_st, sizeof(double) * 12);
    reg("beta", h->beta, sizeof(double) * N * N * ni); reg("beta_f", h->beta_f, sizeof(double) * (N + 1) * nif);
    reg("backoff", h->backoff, sizeof(double) * N * ni); reg("backoff_f", h->backoff_f, sizeof(double) * nif);
    reg("backoff_x", h->backoff_x, sizeof(double) * (N + 1) * nx); reg("backoff_u", h->backoff_u, sizeof(double) * N * nu);
    reg("eta", h->eta, sizeof(double) * N * N * ni); reg("eta_f", h->eta_f, sizeof(double) * (N + 1) * nif);
    reg("K", h->K, sizeof(double) * N * (N + 1) * nu * nx); reg("ubg", h->ubg, sizeof(double) * h->mb); reg("lbg", h->lbg, sizeof(double) * h->mb);
    reg("kkt", h->kkt, sizeof(double) * 8); reg("cost_tube", h->cost_tube, sizeof(double));
    reg("nominal_x", h->Xn, sizeof(double) * (N + 1) * nx); reg("nominal_u", h->Un, sizeof(double) * N * nu); reg("x_meas", h->xmeas, sizeof(double) * nx); reg("u0", h->u0, sizeof(double) * nu);
    reg("A", h->A, sizeof(double) * N * nx * nx); reg("Bm", h->Bm, sizeof(double) * N * nx * nu); reg("c", h->c, sizeof(double) * N * nx);
    reg("g", h->g, sizeof(double) * N * ni); reg("gN", h->gN, sizeof(double) * nif); reg("q", h->q, sizeof(double) * h->n); reg("pin_dual", h->pin_dual, sizeof(double) * nx);
    reg("chain_times", h->chain_times, sizeof(unsigned long long) * 4); reg("primal_infeasibility", h->pinf, sizeof(double)); reg("qp_stats", h->qpstat, sizeof(int) * 16); reg("x0_arg", h->x0arg, sizeof(double) * nx);
#ifdef QP_DIAG_SPAN
    if (dalloc(h->owned, &h->qp_diag, B * 64)) return nullptr;
    reg("qp_diag", h->qp_diag, sizeof(int) * 64);
#endif
    return h;
}

extern "C" void slsqp_destroy(slsqp_handle *h) {
    if (!h) return;
    hipSetDevice(h->dev);
    hipStreamSynchronize(h->st);
    free_all(h->log_owned);
    free_all(h->owned);
    if (h->stage) hipFree(h->stage);
    if (h->cl_W) hipFree(h->cl_W);
    if (h->qplog) hipFree(h->qplog);
    if (h->chain_times_host) hipHostFree(h->chain_times_host);
    for (auto &e : h->ev) hipEventDestroy(e);
    for (auto &e : h->tl) hipEventDestroy(e);
    for (auto &e : h->kev) hipEventDestroy(e);
    hipStreamDestroy(h->st);
    delete h;
}

extern "C" void *slsqp_stream(slsqp_handle *h) { return (void *)h->st; }
extern "C" int slsqp_sync(slsqp_handle *h) { hipSetDevice(h->dev); HIPCHK(hipStreamSynchronize(h->st)); return 0; }

static int put(slsqp_handle *h, void *dst, const void *src, size_t bytes, int loc) {
    if (!src) return 0;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, loc == SLSQP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->st));
    if (loc == SLSQP_HOST) HIPCHK(hipStreamSynchronize(h->st));  // caller's buffer may be reused on return
    return 0;
}

// Device staging area for host-buffer (SLSQP_HOST) entry points: grown on demand, reused, filled with async copies on the handle's stream --
// no hipMalloc / null-stream hipMemcpy per call (those serialise against every other handle's stream).
static double *stage_buf(slsqp_handle *h, size_t bytes) {
    if (bytes > h->stage_bytes) {
        hipStreamSynchronize(h->st);
        if (h->stage) { hipFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
        if (hipMalloc((void **)&h->stage, bytes + 64) != hipSuccess) { h->stage = nullptr; fail("hipMalloc (staging buffer)"); return nullptr; }
        h->stage_bytes = bytes;
    }
    return h->stage;
}

// E (N+1, nx, nw), nw <= nx, into the handle's (N+1, nx, nx) buffer: Phi_x[j,j] = E_j (fast_SLS_jit.py:104-108) with nw < nx (dyn/LTV.py:17-32
// allows any nw) propagates as an nx x nx block whose last nx - nw columns are and stay zero -- every row norm, hence beta, the back-offs and
// cost_tube, is the one of the nx x nw block.
__global__ void k_pad_cols(int rows, int nw, int nx, const double *src, double *dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * nx; i += gridDim.x * blockDim.x) { const int r = i / nx, c = i % nx; dst[i] = c < nw ? src[r * nw + c] : 0.0; }
}
static int put_E(slsqp_handle *h, const double *E, int loc) {
    if (!E) return 0;
    const slsqp_dims &d = h->d;
    const size_t rows = (size_t)(d.N + 1) * d.nx;
    if (d.nw == d.nx) return put(h, h->E, E, sizeof(double) * rows * d.nx, loc);
    const double *src = E;
    if (loc == SLSQP_HOST) {
        double *tmp = stage_buf(h, sizeof(double) * rows * d.nw);
        if (!tmp) return -1;
        HIPCHK(hipMemcpyAsync(tmp, E, sizeof(double) * rows * d.nw, hipMemcpyHostToDevice, h->st));
        src = tmp;
    }
    hipLaunchKernelGGL(k_pad_cols, dim3(64), dim3(256), 0, h->st, (int)rows, d.nw, d.nx, src, h->E);
    HIPCHK(hipGetLastError());
    if (loc == SLSQP_HOST) HIPCHK(hipStreamSynchronize(h->st));
    return 0;
}

static bool is_diag(const double *M, int n) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) if (i != j && M[i * n + j] != 0.0) return false;
    return true;
}

extern "C" int slsqp_set_costs(slsqp_handle *h, const double *Q, const double *R, const double *Qf, const double *Q_reg, const double *R_reg,
                               const double *Q_reg_f) {
    hipSetDevice(h->dev);
    const int nx = h->d.nx, nu = h->d.nu;
    if (!is_diag(Q, nx) || !is_diag(R, nu) || !is_diag(Qf, nx) || !is_diag(Q_reg, nx) || !is_diag(R_reg, nu) || !is_diag(Q_reg_f, nx))
        return fail("Q,R,Qf,Q_reg,R_reg,Q_reg_f must be diagonal (HIP path); dense weights are not supported");
    std::vector<double> v;
    auto push = [&](const double *M, int n) { for (int i = 0; i < n; i++) v.push_back(M[i * n + i]); };
    push(Q, nx); push(R, nu); push(Qf, nx); push(Q_reg, nx); push(R_reg, nu); push(Q_reg_f, nx);
    for (int i = 0; i < nx + nu + nx; i++) if (!(v[i] > 0.0)) return fail("Q,R,Qf must be positive definite");
    HIPCHK(hipMemcpy(h->cst, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    h->have_costs = true;
    return 0;
}

extern "C" int slsqp_set_constraints(slsqp_handle *h, const double *G, const double *Gf, const double *gf) {
    hipSetDevice(h->dev);
    const int nx = h->d.nx, nz = h->nz, ni = h->d.ni, nif = h->d.ni_f;
    bool box = (ni == 2 * nz) && (nif == 2 * nx);
    for (int i = 0; box && i < 2 * nz; i++) for (int j = 0; j < nz; j++) {
        const double want = (i % nz == j) ? (i < nz ? 1.0 : -1.0) : 0.0;
        if (G[i * nz + j] != want) { box = false; break; }
    }
    for (int i = 0; box && i < 2 * nx; i++) for (int j = 0; j < nx; j++) {
        const double want = (i % nx == j) ? (i < nx ? 1.0 : -1.0) : 0.0;
        if (Gf[i * nx + j] != want) { box = false; break; }
    }
    // general G (fast_SLS_jit.py:76-79, 138-170; Pendulum.replace_constraints): the SLS sweep takes it (k_sweep_gen); the QP solver eliminates box
    // constraints through a diagonal Pi and does not -- slsqp_solve / slsqp_qp_* / slsqp_cl_* refuse such a handle
    h->general_G = !box;
    HIPCHK(hipMemcpy(h->Gd, G, sizeof(double) * (size_t)ni * nz, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->Gfd, Gf, sizeof(double) * (size_t)nif * nx, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->gf_raw, gf, sizeof(double) * nif, hipMemcpyHostToDevice));
    h->have_cons = true;
    return 0;
}

extern "C" int slsqp_update_dynamics(slsqp_handle *h, const double *A, const double *Bm, const double *E, const double *g, const double *g_N,
                                     const double *c, int loc) {
    hipSetDevice(h->dev);
    const slsqp_dims &d = h->d;
    const size_t B = h->B;
    if (!A || !Bm || !g || !g_N || !c) return fail("A, B, g, g_N, c are required");
    if (put(h, h->A, A, sizeof(double) * B * d.N * d.nx * d.nx, loc)) return -1;
    if (put(h, h->Bm, Bm, sizeof(double) * B * d.N * d.nx * d.nu, loc)) return -1;
    if (put_E(h, E, loc)) return -1;
    if (put(h, h->g, g, sizeof(double) * B * d.N * d.ni, loc)) return -1;
    if (put(h, h->gN, g_N, sizeof(double) * B * d.ni_f, loc)) return -1;
    if (put(h, h->c, c, sizeof(double) * B * d.N * d.nx, loc)) return -1;
    BoundsArgs a{h->B, d.N, d.nx, d.ni, d.ni_f, h->g, h->gN, h->c, h->ubg, h->lbg, 1e-10, nullptr};
    hipLaunchKernelGGL(k_set_bounds, dim3(1024), dim3(256), 0, h->st, a);
    HIPCHK(hipGetLastError());
    h->have_dyn = true;
    return 0;
}

extern "C" int slsqp_update_linear_cost(slsqp_handle *h, const double *q, int loc) {
    hipSetDevice(h->dev);
    return put(h, h->q, q, sizeof(double) * (size_t)h->B * h->n, loc);
}

// ---- small mask kernels ------------------------------------------------------------------------------------
__global__ void k_fill_int(int *p, int v, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void k_fill_doubles(double *p, double v, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void k_negate(const double *x, double *y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = -x[i]; }
__global__ void k_finish(int B, int rti, const int *alive, const int *infeas, int *success, const int *active, int *pending_reset) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (active && !active[b]) { success[b] = 0; return; }       // not part of this call
    if (rti) success[b] = (!infeas[b]) || success[b];           // fast_SLS_jit.py:295
    else if (alive[b] || infeas[b]) { success[b] = 0; pending_reset[b] = 1; }   // hit MAX_ITER or infeasible (:304, :312) -> _finish_failure
}
// _finish_failure (fast_SLS_jit.py:334-341) resets the solver AFTER the result dict has been built: the reset (eta, eta_f,
// iteration_number -> 0; bounds and linear cost are rewritten by the caller's next update anyway) is applied at the top of the
// instance's next solve, so that slsqp_get still returns the failed call's arrays.
__global__ void k_apply_pending_reset(int B, int *pending, const int *active, int *itnum, double *eta, size_t neta, double *eta_f, size_t netaf, int *stale) {
    const int b = blockIdx.x;
    if (!pending[b] || (active && !active[b])) return;
    for (size_t o = threadIdx.x; o < neta; o += blockDim.x) eta[(size_t)b * neta + o] = 0.0;
    for (size_t o = threadIdx.x; o < netaf; o += blockDim.x) eta_f[(size_t)b * netaf + o] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) { itnum[b] = 0; pending[b] = 0; stale[b] &= ~16; }
}
// The elementwise work between the first QP of a fast-SLS iteration and its sweep in ONE launch (one workgroup per instance), in the reference's order:
//   a failed QP drops the instance (forward_solve -> False, fast_SLS_jit.py:461-464);
//   evaluate_dual_eta (:475-487) -- in the first iteration of a solve beta = eps everywhere, eta[k,j] = mu_k / (2 sqrt(eps)) for every j <= k and only
//     column 0 is written (all the shared Riccati recursion reads; slsqp_get broadcasts it on demand);
//   check_convergence_socp (:581-600, its state persists across calls: quirk q5);
//   the mask of the instances that go on to Riccati + tightening (_step :318-326);
//   first iteration only: instances that are NOT swept but whose beta still holds an earlier solve's values get initialize_backoff's eps (the solve
//     start no longer writes beta for everybody: 1.1 GB per 4096 rocket instances).
// Five launches before; a pendulum closed-loop step is launch-bound and 30 of its ~120 launches were these.
struct AfterQpArgs {
    int B, rti, first_iter;
    const int *status, *active;
    int *alive, *infeas, *mask, *success, *itnum, *counter, *stale, *conv;
    EtaArgs ea; ConvArgs ca;
    double *beta_w, *beta_f_w;     // the same arrays as ea.beta / ea.beta_f, writable (the beta repair)
};
__global__ __launch_bounds__(256) void k_after_qp(AfterQpArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ int s_alive, s_m, s_stale;
    if (tid == 0) {
        int al = a.alive[b];
        if (al) { const int st = a.status[b]; if (!(st == 0 || st == 4)) { al = 0; a.alive[b] = 0; a.infeas[b] = 1; } }
        s_alive = al;
    }
    __syncthreads();
    const int alive = s_alive;
    if (alive) {       // ---- evaluate_dual_eta
        const EtaArgs &e = a.ea;
        const int SR = e.NX + e.NI, mb = e.N * SR + e.NIF;
        const double *du = e.dual + (size_t)b * mb;
        double *et = e.eta + (size_t)b * e.N * e.N * e.NI, *ef = e.eta_f + (size_t)b * (e.N + 1) * e.NIF;
        if (e.first_iter) {
            const double s0 = 2.0 * sqrt(e.eps);
            for (int o = tid; o < e.N * e.NI; o += blockDim.x) { const int i = o % e.NI, k = o / e.NI; et[((size_t)k * e.N) * e.NI + i] = du[k * SR + e.NX + i] / s0; }
            for (int o = tid; o < e.NIF; o += blockDim.x) ef[o] = du[e.N * SR + o] / s0;
        } else {
            const double *be = e.beta + (size_t)b * e.N * e.N * e.NI, *bf = e.beta_f + (size_t)b * (e.N + 1) * e.NIF;
            for (int o = tid; o < e.N * e.N * e.NI; o += blockDim.x) {
                const int i = o % e.NI, j = (o / e.NI) % e.N, k = o / (e.NI * e.N);
                if (j <= k) et[o] = du[k * SR + e.NX + i] / (2.0 * sqrt(fmax(be[o], e.eps)));
            }
            for (int o = tid; o < (e.N + 1) * e.NIF; o += blockDim.x) ef[o] = du[e.N * SR + (o % e.NIF)] / (2.0 * sqrt(fmax(bf[o], e.eps)));
        }
    }
    int cv = 0;
    if (tid < 64) {    // ---- check_convergence_socp (first wave)
        const ConvArgs &c = a.ca;
        if (alive) {
            const double *p = c.primal + (size_t)b * c.n;
            double *q = c.prev + (size_t)b * c.n;
            double d = 0.0;
            for (int o = tid; o < c.n; o += 64) { d = fmax(d, fabs(p[o] - q[o])); q[o] = p[o]; }
            d = wla::wave_max(d);
            if (tid == 0) { cv = (c.has_prev[b] && d <= c.tol) ? 1 : 0; c.has_prev[b] = 1; }
        }
        if (tid == 0) {
            a.conv[b] = cv;
            int st = a.stale[b];
            if (alive) st = a.ea.first_iter ? ((st & ~1) | 16) : (st & ~(1 | 16));      // state of the eta array: column 0 only / complete
            int m = 0;                                                                    // ---- mask
            if (alive) {
                if (cv) { a.success[b] = 1; if (!a.rti) a.alive[b] = 0; }
                else m = 1;
            }
            a.mask[b] = m;
            if (m) { a.itnum[b] += 1; st = (st & ~(2 | 32)) | 8; if (!a.rti) atomicAdd(a.counter, 1); }
            s_m = m; s_stale = st;
            a.stale[b] = st;
        }
    }
    __syncthreads();
    if (a.first_iter && !((a.active && !a.active[b]) || s_m || !(s_stale & 8))) {      // ---- beta repair
        const EtaArgs &e = a.ea;
        double *be = a.beta_w + (size_t)b * e.N * e.N * e.NI, *bf = a.beta_f_w + (size_t)b * (e.N + 1) * e.NIF;
        for (int o = tid; o < e.N * e.N * e.NI; o += blockDim.x) be[o] = e.eps;
        for (int o = tid; o < (e.N + 1) * e.NIF; o += blockDim.x) bf[o] = e.eps;
        if (tid == 0) a.stale[b] = s_stale & ~8;
    }
}

// The start of a fast-SLS solve in one launch (one workgroup per instance): x_0 pin value, alive mask, the pending solver reset of an instance
// whose last converge-mode solve failed, cleared flags, initialize_backoff.  Replaces seven launches.
struct SolveBeginArgs {
    int B, NX;
    const double *x0;            // (B,NX) device, or NULL when the pin value has already been written
    double *x0val;
    const int *active; int *alive, *infeas, *success, *pending, *itnum, *stale;
    double *eta, *eta_f; size_t neta, netaf;
    InitBackoffArgs ib;          // run = the instances whose back-offs are reset
    const int *skip;             // (B) or NULL: instances that are in the middle of a solve (slsqp_cl_run) and must not be touched at all
};
__global__ __launch_bounds__(256) void k_solve_begin(SolveBeginArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.skip && a.skip[b]) return;
    const int act = a.active ? a.active[b] : 1;
    if (a.x0 && tid < a.NX) a.x0val[(size_t)b * a.NX + tid] = -a.x0[(size_t)b * a.NX + tid];
    if (tid == 0) { a.alive[b] = act; a.infeas[b] = 0; a.success[b] = 0; }
    if (a.pending[b] && act) {       // _finish_failure of the previous call (_finish_failure, fast_SLS_jit.py:334-341)
        for (size_t o = tid; o < a.neta; o += blockDim.x) a.eta[(size_t)b * a.neta + o] = 0.0;
        for (size_t o = tid; o < a.netaf; o += blockDim.x) a.eta_f[(size_t)b * a.netaf + o] = 0.0;
        __syncthreads();
        if (tid == 0) { a.itnum[b] = 0; a.pending[b] = 0; a.stale[b] &= ~16; }
    }
    const InitBackoffArgs &i = a.ib;
    if (i.run && !i.run[b]) return;
    const int NZ = i.NX + i.NU, NI = 2 * NZ, NIF = 2 * i.NX, N = i.N;
    const double sq = sqrt(i.eps);
    for (int o = tid; o < N * NI; o += blockDim.x) i.backoff[(size_t)b * N * NI + o] = N * sq;
    for (int o = tid; o < NIF; o += blockDim.x) i.backoff_f[(size_t)b * NIF + o] = (N + 1) * sq;
    for (int o = tid; o < (N + 1) * i.NX; o += blockDim.x) i.backoff_x[(size_t)b * (N + 1) * i.NX + o] = 0.0;
    for (int o = tid; o < N * i.NU; o += blockDim.x) i.backoff_u[(size_t)b * N * i.NU + o] = 0.0;
}

// k_after_qp for ONE wave (the fused chain kernel below): the same steps in the same order, reductions through the wave instead of LDS.
__device__ __forceinline__ int after_qp_wave(const AfterQpArgs &a, int b, int lane) {
    int alive = a.alive[b];
    if (alive) { const int st = a.status[b]; if (!(st == 0 || st == 4)) { alive = 0; if (lane == 0) { a.alive[b] = 0; a.infeas[b] = 1; } } }
    const EtaArgs &e = a.ea;
    if (alive) {       // ---- evaluate_dual_eta
        const int SR = e.NX + e.NI, mb = e.N * SR + e.NIF;
        const double *du = e.dual + (size_t)b * mb;
        double *et = e.eta + (size_t)b * e.N * e.N * e.NI, *ef = e.eta_f + (size_t)b * (e.N + 1) * e.NIF;
        if (e.first_iter) {
            const double s0 = 2.0 * sqrt(e.eps);
            for (int o = lane; o < e.N * e.NI; o += 64) { const int i = o % e.NI, k = o / e.NI; et[((size_t)k * e.N) * e.NI + i] = du[k * SR + e.NX + i] / s0; }
            for (int o = lane; o < e.NIF; o += 64) ef[o] = du[e.N * SR + o] / s0;
        } else {
            const double *be = e.beta + (size_t)b * e.N * e.N * e.NI, *bf = e.beta_f + (size_t)b * (e.N + 1) * e.NIF;
            for (int o = lane; o < e.N * e.N * e.NI; o += 64) {
                const int i = o % e.NI, j = (o / e.NI) % e.N, k = o / (e.NI * e.N);
                if (j <= k) et[o] = du[k * SR + e.NX + i] / (2.0 * sqrt(fmax(be[o], e.eps)));
            }
            for (int o = lane; o < (e.N + 1) * e.NIF; o += 64) ef[o] = du[e.N * SR + (o % e.NIF)] / (2.0 * sqrt(fmax(bf[o], e.eps)));
        }
    }
    int cv = 0;
    const ConvArgs &c = a.ca;
    if (alive) {       // ---- check_convergence_socp
        const double *p = c.primal + (size_t)b * c.n;
        double *q = c.prev + (size_t)b * c.n;
        double d = 0.0;
        for (int o = lane; o < c.n; o += 64) { d = fmax(d, fabs(p[o] - q[o])); q[o] = p[o]; }
        d = wla::wave_max(d);
        cv = (c.has_prev[b] && d <= c.tol) ? 1 : 0;
    }
    int st = a.stale[b];
    if (alive) st = e.first_iter ? ((st & ~1) | 16) : (st & ~(1 | 16));
    int m = 0;                                                                        // ---- mask
    if (alive && !cv) m = 1;
    if (m) st = (st & ~(2 | 32)) | 8;
    const bool repair = a.first_iter && !((a.active && !a.active[b]) || m || !(st & 8));
    wla::wsync_mem();                  // every lane has read has_prev / stale / alive before lane 0 rewrites them
    if (lane == 0) {
        if (alive) c.has_prev[b] = 1;
        a.conv[b] = cv;
        if (alive && cv) { a.success[b] = 1; if (!a.rti) a.alive[b] = 0; }
        a.mask[b] = m;
        if (m) { a.itnum[b] += 1; if (!a.rti) atomicAdd(a.counter, 1); }
        a.stale[b] = repair ? (st & ~8) : st;
    }
    if (repair) {      // ---- beta repair
        double *be = a.beta_w + (size_t)b * e.N * e.N * e.NI, *bf = a.beta_f_w + (size_t)b * (e.N + 1) * e.NIF;
        for (int o = lane; o < e.N * e.N * e.NI; o += 64) be[o] = e.eps;
        for (int o = lane; o < (e.N + 1) * e.NIF; o += 64) bf[o] = e.eps;
    }
    return m;
}

// One RTI fast-SLS solve (fast_SLS.solve with rti_steps = 1, the rocket script's setting: QP -> eta -> Riccati / propagation / back-offs ->
// tightened bounds -> QP) of one instance by ONE wave in ONE launch.  The separate launches (k_qp_solve, k_after_qp, k_sweep_ric1, k_sweep_prop,
// k_tighten, k_qp_solve) put a batch-wide barrier behind each QP: every launch lasts as long as its slowest instance (a failed warm attempt
// followed by the interior point: 50-90 block solves against a mean of 10), so the step pays max(QP #1) + max(QP #2).  Here an instance that is
// slow in one QP only delays itself and the step pays max over the instances of their own sums.  Same device functions, same arithmetic, same bits.
struct ChainArgs {
    QpArgs q1, q2;
    AfterQpArgs aq;
    SweepSharedArgs sw;
    TightenArgs ta;
    const int *active; int *success; const int *infeas;      // k_finish
    unsigned long long *times;      // (B,4) or NULL: wall_clock64 ticks (100 MHz) of QP #1, the sweep part, QP #2 of each instance
    int max_ticks;
    // slsqp_cl_run (all NULL / 0 otherwise): instances advance through their MPC steps independently
    int *lag;                       // (B) 0: the instance starts its chain in this launch; 1 / 2: it was suspended inside QP #1 / QP #2 and resumes there
    const int *runm;                // (B) 0: the instance takes no part in this launch (it has finished all its steps)
    int *done;                      // (B) out: 1 = the chain of the instance ended in this launch
    unsigned long long *t0word;     // start time of the launch (first wave to arrive writes it), zeroed by the host before the launch
    unsigned long long budget;      // wall-clock ticks (100 MHz) after the start at which unfinished solves suspend themselves
    unsigned *fin_count; unsigned cut_count;      // chains of this launch that have ended (device counter, zeroed by the host); solves suspend once it reaches cut_count
    int *qplog; const int *stepno; int log_steps;      // (B, log_steps, 16) per-step copy of the instance's qp_stats, entry stepno[b]
};
template <int NX, int NU>
__device__ __forceinline__ int rti_chain_dev(const ChainArgs &c, int b, int lane, double *sm) {
    unsigned long long t0 = wall_clock64(), t1 = t0, t2 = t0, deadline = ~0ULL;
    int lg = 0;
    if (c.lag) {
        lg = __builtin_amdgcn_readfirstlane(c.lag[b]);
        unsigned long long first = 0ULL;
        if (lane == 0) { first = atomicCAS(c.t0word, 0ULL, t0); if (first == 0ULL) first = t0; }
        first = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(first >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)first);      // (the builtin returns a signed int: no sign extension)
        deadline = first + c.budget;
        if (lane == 0) c.done[b] = 0;
    }
    int fin = 1;
#pragma unroll 1
    for (int pass = (lg == 2 ? 1 : 0); pass < 2; pass++) {
        asm volatile("" : "+v"(lane));
        fin = __builtin_amdgcn_readfirstlane(qp_solve_dev<NX, NU, false>(pass == 0 ? c.q1 : c.q2, b, lane, sm, c.max_ticks, (lg == pass + 1) ? 1 : 0, deadline, c.lag ? c.fin_count : nullptr, c.cut_count));
        wla::wsync_mem();
        if (!fin) { if (lane == 0) c.lag[b] = pass + 1; break; }      // suspended at the deadline: the next launch resumes this solve
        if (pass == 1) break;
        t1 = wall_clock64();
        asm volatile("" : "+v"(lane));
        const int m = __builtin_amdgcn_readfirstlane(after_qp_wave(c.aq, b, lane));      // (wave-uniform by construction; said so to the compiler)
        wla::wsync_mem();
        if (m) {       // Riccati + tightening for this instance (fast_SLS_jit.py:321-322)
            sweep_ric1_dev<NX, NU>(c.sw, b, lane, sm);
            wla::wsync_mem();
#pragma unroll 1
            for (int j = 0; j <= c.sw.s.N; j += 2) {      // two disturbance columns per pass
                asm volatile("" : "+v"(lane));
                sweep_prop_dev<NX, NU>(c.sw, b, j, min(2, c.sw.s.N + 1 - j), lane, sm);
                wla::wsync();
            }
            wla::wsync_mem();
            tighten_dev(c.ta, b, lane, 64);
            wla::wsync_mem();
        }
        t2 = wall_clock64();
    }
    if (!fin) return 0;
    if (lane == 0) {
        if (c.active && !c.active[b]) c.success[b] = 0;
        else c.success[b] = (!c.infeas[b]) || c.success[b];           // fast_SLS_jit.py:295 (k_finish, RTI)
        if (c.times) { const unsigned long long t3 = wall_clock64(); unsigned long long *t = c.times + (size_t)b * 4; t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t3 - t0; }
        if (c.lag) { c.lag[b] = 0; c.done[b] = 1; atomicAdd(c.fin_count, 1u); }
    }
    if (c.qplog && lane < 16) c.qplog[((size_t)b * c.log_steps + min(c.stepno[b], c.log_steps - 1)) * 16 + lane] = c.q1.qpstat[(size_t)b * 16 + lane];
    return 1;
}
template <int NX, int NU>
__global__ __launch_bounds__(64, QP_PERSIST_WAVES_PER_SIMD) void k_rti_chain(ChainArgs c) {
    int b = blockIdx.x, lane = threadIdx.x;
    if (b >= c.q1.B) return;
    if (c.runm && !c.runm[b]) return;
    extern __shared__ double sm[];
    rti_chain_dev<NX, NU>(c, b, lane, sm);
}

// ---- the whole closed loop as ONE persistent launch (slsqp_cl_run, opts.cl_persistent) ------------------------------------------------
// A wave takes an instance from a device-side FIFO, runs ONE complete MPC step of it -- warm-start shift, solver reset, linearisation, x0 pin,
// solve start, the RTI chain (QP #1 -> eta -> Riccati / propagation -> tightened bounds -> QP #2), nominal update, primal infeasibility, log
// entry, plant step with the disturbance sample of that step -- and puts the instance back while it has steps left.  Every step of every
// instance is the same sequence of device functions the launches of slsqp_cl_step run (same arithmetic, same bits); what is gone is every
// barrier between instances: no wave slot waits for a round, a deadline or the slowest instance of a batch, and the FIFO order keeps the
// instances' step counters together so the run has no long tail.  The queue: slots[cap] (0 = empty, else instance + 1), tail / head
// tickets, `avail` = published items not yet claimed.  A wave that finds avail <= 0 EXITS (an instance is always either in the queue or
// held by a running wave, which will push and then pop again; so nothing is stranded); the only waits are for the slot of a ticket already
// taken to be published / cleared by a wave that is a few instructions away from doing so, and they are bounded (err flag instead of a hang).
// Hand-over between waves on different XCDs (own L2 each): agent-scope release before the push, acquire after the pop.
struct ClQueue { int *slots; unsigned mask; unsigned *head, *tail; int *avail, *err; };
__device__ __forceinline__ int clq_pop(const ClQueue &q, int lane) {
    int v = -1;
    if (lane == 0) {
        const int av = atomicSub(q.avail, 1);
        if (av <= 0) atomicAdd(q.avail, 1);
        else {
            const unsigned t = atomicAdd(q.head, 1u);
            int *slot = q.slots + (t & q.mask);
            int got = 0;
            for (int spin = 0; spin < (1 << 22) && !got; spin++) {
                got = __hip_atomic_exchange(slot, 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (!got) __builtin_amdgcn_s_sleep(4);
            }
            if (got) v = got - 1; else atomicExch(q.err, 1);
        }
    }
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void clq_push(const ClQueue &q, int b, int lane) {
    if (lane == 0) {
        const unsigned p = atomicAdd(q.tail, 1u);
        int *slot = q.slots + (p & q.mask);
        bool ok = false;
        for (int spin = 0; spin < (1 << 22) && !ok; spin++) {
            int expected = 0;
            ok = __hip_atomic_compare_exchange_strong(slot, &expected, b + 1, __ATOMIC_RELEASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!ok) __builtin_amdgcn_s_sleep(4);
        }
        if (ok) atomicAdd(q.avail, 1); else atomicExch(q.err, 2);
    }
}
__global__ void k_clq_init(int B, ClQueue q) {
    const unsigned cap = q.mask + 1u;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) q.slots[i] = (i < (unsigned)B) ? (int)i + 1 : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *q.head = 0u; *q.tail = (unsigned)B; *q.avail = B; *q.err = 0; }
}
// k_solve_begin for one wave
__device__ __forceinline__ void solve_begin_wave(const SolveBeginArgs &a, int b, int lane) {
    const int act = a.active ? a.active[b] : 1;
    if (a.x0 && lane < a.NX) a.x0val[(size_t)b * a.NX + lane] = -a.x0[(size_t)b * a.NX + lane];
    const int pend = a.pending[b];
    wla::wsync_mem();
    if (lane == 0) { a.alive[b] = act; a.infeas[b] = 0; a.success[b] = 0; }
    if (pend && act) {       // _finish_failure of the previous call (fast_SLS_jit.py:334-341)
        for (size_t o = lane; o < a.neta; o += 64) a.eta[(size_t)b * a.neta + o] = 0.0;
        for (size_t o = lane; o < a.netaf; o += 64) a.eta_f[(size_t)b * a.netaf + o] = 0.0;
        const int st = a.stale[b];
        wla::wsync_mem();
        if (lane == 0) { a.itnum[b] = 0; a.pending[b] = 0; a.stale[b] = st & ~16; }
    }
    const InitBackoffArgs &i = a.ib;
    if (i.run && !i.run[b]) return;
    const int NZ = i.NX + i.NU, NI = 2 * NZ, NIF = 2 * i.NX, N = i.N;
    const double sq = sqrt(i.eps);
    for (int o = lane; o < N * NI; o += 64) i.backoff[(size_t)b * N * NI + o] = N * sq;
    for (int o = lane; o < NIF; o += 64) i.backoff_f[(size_t)b * NIF + o] = (N + 1) * sq;
    for (int o = lane; o < (N + 1) * i.NX; o += 64) i.backoff_x[(size_t)b * (N + 1) * i.NX + o] = 0.0;
    for (int o = lane; o < N * i.NU; o += 64) i.backoff_u[(size_t)b * N * i.NU + o] = 0.0;
}
struct LoopArgs {
    ChainArgs c; ClArgs cl; LinArgs lin; BoundsArgs ba; SolveBeginArgs sb; ClLogArgs lg; ScpArgs sa;
    int steps, n, have_log, fence, keep_laggards;
    int *stepno; double *call_ids, *q; int *stale, *itnum, *pending, *scp_active;
    const double *W_all; double *pinf;
    ClQueue Q;
    unsigned long long *busy, *t_begin;      // busy[0] sum over the MPC steps of the time a wave spent on them (100 MHz ticks), [2] MPC steps run; t_begin (B): start of the instance's current step
};
// the two halves of an MPC step around the RTI chain, as functions of their own: what they keep in registers (dual numbers of the linearisation, the
// plant's RK4 stages) stays out of the register allocation of the QP loops, and nothing of theirs is live across the chain
template <int MODEL>
__device__ CLW_FN void cl_step_begin(const LoopArgs &L, int b, int lane) {
    constexpr int NX = dyn::Dims<MODEL>::NX;
    const int s = L.stepno[b];
    if (lane == 0) L.t_begin[b] = wall_clock64();
#ifdef CL_LOOP_STAMP
    unsigned long long ts_ = wall_clock64();
#define CLSTAMP(i) do { wla::wsync_mem(); const unsigned long long t_ = wall_clock64(); if (lane == 0) atomicAdd(L.busy + (i), t_ - ts_); ts_ = t_; } while (0)
#else
#define CLSTAMP(i) do {} while (0)
#endif
    // reset_warm_start (shift + solver reset) past the first step, call id, flags
    if (s > 0) {
        cl_shift_wave<MODEL>(L.cl, b, lane);
        for (int o = lane; o < L.n; o += 64) L.q[(size_t)b * L.n + o] = 0.0;
        const int st = L.stale[b];
        wla::wsync_mem();
        if (lane == 0) { L.stale[b] = (st & 8) | 3; L.itnum[b] = 0; L.pending[b] = 0; }
    }
    if (lane == 0) { L.call_ids[b] += 1.0; L.sa.scp_success[b] = 0; L.sa.scp_iters[b] = 0; }
    wla::wsync_mem();
    CLSTAMP(4);
    lin_wave<MODEL>(L.lin, L.ba, b, lane);
    CLSTAMP(5);
    if (lane < NX) L.cl.x0arg[(size_t)b * NX + lane] = L.cl.Xn[(size_t)b * (L.cl.N + 1) * NX + lane] - L.cl.xmeas[(size_t)b * NX + lane];
    wla::wsync_mem();
    solve_begin_wave(L.sb, b, lane);
    wla::wsync_mem();
    CLSTAMP(6);
}
// nominal += delta, primal infeasibility, log entry, plant + noise, step counter; returns the instance's new step count while it has steps left, else 0
template <int MODEL>
__device__ CLW_FN int cl_step_end(const LoopArgs &L, int b, int lane) {
    constexpr int NX = dyn::Dims<MODEL>::NX;
    const int s = L.stepno[b];
#ifdef CL_LOOP_STAMP
    unsigned long long ts_ = wall_clock64();
#endif
    if (lane == 0) L.scp_active[b] = 1;
    wla::wsync_mem();
    cl_scp_update_wave(L.cl, L.sa, b, lane);
    wla::wsync_mem();
    CLSTAMP(7);
    if (L.sa.updated[b]) cl_infeas_wave<MODEL>(L.cl, L.pinf, b, lane);
    wla::wsync_mem();
    CLSTAMP(8);
    if (L.have_log) {
        const int per = 2 * ((L.lg.N + 1) * L.lg.NX + L.lg.N * L.lg.NU);
        for (int o = lane; o < per; o += 64) cl_log_item(L.lg, b, o);
    }
    CLSTAMP(9);
    if (lane == 0) {
        cl_plant_one<MODEL>(L.cl, b, L.W_all ? L.W_all + (size_t)s * L.cl.B * NX : nullptr);
        L.stepno[b] = s + 1;
        if (L.busy) { atomicAdd(L.busy, wall_clock64() - L.t_begin[b]); atomicAdd(L.busy + 2, 1ULL); }
    }
    wla::wsync_mem();
    CLSTAMP(10);
    return (s + 1 < L.steps) ? s + 1 : 0;
}
template <int MODEL>
__global__ __launch_bounds__(64, QP_PERSIST_WAVES_PER_SIMD) void k_cl_loop(LoopArgs L) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    int lane = threadIdx.x;
    extern __shared__ double sm[];
    int b = -1;
#ifdef CL_LOOP_STAMP
    const unsigned long long tw0_ = wall_clock64();
    if (lane == 0) atomicMin(L.busy + 14, tw0_);
#endif
#pragma unroll 1
    for (;;) {
        asm volatile("" : "+v"(lane));
        if (b < 0) {
#ifdef CL_LOOP_STAMP
            const unsigned long long tp_ = wall_clock64();
#endif
            b = clq_pop(L.Q, lane);
            if (b < 0) break;
            if (L.fence & 1) __threadfence();      // acquire: what the wave that ran this instance's previous step wrote (possibly through another XCD's L2)
#ifdef CL_LOOP_STAMP
            if (lane == 0) atomicAdd(L.busy + 11, wall_clock64() - tp_);
#endif
        }
        cl_step_begin<MODEL>(L, b, lane);
        b = __builtin_amdgcn_readfirstlane(b);      // (across a call the compiler may park it in a vector register)
        asm volatile("" : "+v"(lane));
        rti_chain_dev<NX, NU>(L.c, b, lane, sm);
        wla::wsync_mem();
        asm volatile("" : "+v"(lane));
        const int next = __builtin_amdgcn_readfirstlane(cl_step_end<MODEL>(L, b, lane));
        b = __builtin_amdgcn_readfirstlane(b);
        if (!next) { b = -1; continue; }
        // An instance that is behind the batch's mean progress keeps its wave and goes straight on (a few instances are slow in MANY of their
        // steps: queueing after each of them they would finish long after the others, with the GPU nearly empty); the others queue up, so the
        // waves are shared fairly among the instances that are level
        const unsigned long long done_steps = __hip_atomic_load(L.busy + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int behind = __builtin_amdgcn_readfirstlane(((unsigned long long)next * (unsigned long long)L.cl.B < done_steps) ? 1 : 0);
        if (behind && L.keep_laggards) continue;
#ifdef CL_LOOP_STAMP
        const unsigned long long tq_ = wall_clock64();
#endif
        if (L.fence & 2) __threadfence();      // release: the next step of this instance may run anywhere
        clq_push(L.Q, b, lane);
        b = -1;
#ifdef CL_LOOP_STAMP
        if (lane == 0) atomicAdd(L.busy + 12, wall_clock64() - tq_);
#endif
    }
#ifdef CL_LOOP_STAMP
    if (lane == 0) { const unsigned long long te_ = wall_clock64(); atomicAdd(L.busy + 13, te_ - tw0_); atomicMax(L.busy + 15, te_); atomicAdd(L.busy + 1, 1ULL); }
#endif
}

// ---- masked pieces of a closed-loop round (slsqp_cl_run) ----
// cl_stage[b]: what the instance does in this round -- begin = it starts its next MPC step (shift, reset, linearise, x0 pin, solve start), runm = it runs
// the chain (begins or resumes), from its step counter and lag state
__global__ void k_cl_round_masks(int B, int steps, const int *stepno, const int *lag, int *begin, int *runm, int *skip_begin, int *done, int *n_unfinished) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int unfinished = stepno[b] < steps;
    const int bg = unfinished && lag[b] == 0;
    begin[b] = bg; runm[b] = unfinished; skip_begin[b] = !bg;
    done[b] = 0;       // (set by the chain kernel for the instances whose chain ends in this round; instances that take no part must not keep an old 1)
    if (unfinished) atomicAdd(n_unfinished, 1);
}
// reset_solver_to_zeros of slsqp_reset for the instances that begin a step (the others are in the middle of theirs)
__global__ void k_cl_reset_masked(int B, int n, const int *begin, const int *stepno, int *stale, int *itnum, int *pending, double *q) {
    const int b = blockIdx.x;
    if (!begin[b] || stepno[b] == 0) return;
    for (int o = threadIdx.x; o < n; o += blockDim.x) q[(size_t)b * n + o] = 0.0;
    if (threadIdx.x == 0) { stale[b] = (stale[b] & 8) | 3; itnum[b] = 0; pending[b] = 0; }
}
__global__ void k_cl_advance(int B, const int *done, int *stepno, double *call_ids, const int *begin) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (done[b]) stepno[b] += 1;
}
__global__ void k_cl_bump_call(int B, const int *begin, double *call_ids) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && begin[b]) call_ids[b] += 1.0;
}

__global__ void k_reset_stale(int *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (p[i] & 8) | 3; }   // eta, K: zero on demand; beta keeps its state
// instances whose `bit` is set in stale[]: zero their slice of arr1 (and arr2), then clear the bit (last use decides: clear_bit)
__global__ void k_zero_stale(int *stale, int bit, double *arr1, size_t n1, double *arr2, size_t n2) {
    const int b = blockIdx.x;
    if (!(stale[b] & bit)) return;
    for (size_t o = threadIdx.x; o < n1; o += blockDim.x) arr1[(size_t)b * n1 + o] = 0.0;
    if (arr2) for (size_t o = threadIdx.x; o < n2; o += blockDim.x) arr2[(size_t)b * n2 + o] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) stale[b] &= ~bit;
}
__global__ void k_copy_int(const int *src, int *dst, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }
__global__ void k_split_lu(int B, int mb, int nx, const double *l, const double *u, double *lbg, double *ubg, double *x0val) {
    const int m = mb + nx;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < (size_t)B * m; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = idx / m, r = idx % m;
        if (r < mb) { lbg[(size_t)b * mb + r] = l[idx]; ubg[(size_t)b * mb + r] = u[idx]; }
        else x0val[(size_t)b * nx + (r - mb)] = 0.5 * (l[idx] + u[idx]);
    }
}
__global__ void k_gather_csc(int B, int cnt, int nnz, const int *map, const double *Ax, double *dst) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < (size_t)B * cnt; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = idx / cnt, o = idx % cnt;
        dst[idx] = Ax[(size_t)b * nnz + map[o]];
    }
}
__global__ void k_join_y(int B, int mb, int nx, const double *dual, const double *pin, double *y) {
    const int m = mb + nx;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < (size_t)B * m; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = idx / m, r = idx % m;
        y[idx] = r < mb ? dual[(size_t)b * mb + r] : pin[(size_t)b * nx + (r - mb)];
    }
}

// ---- kernel dispatch ---------------------------------------------------------------------------------------
// ticks a QP solve may take: 1 (start) + 2 per interior-point iteration + (1 + n_refine) per polish round, after at most warm_rounds + as_rounds
// active-set rounds of the attempts that precede the interior point
static int qp_max_ticks(const QpArgs &a, int max_iter) { return 1 + 2 * max_iter + 3 + 2 * 10 + 8 + 2 + 16 + 2 * (a.warm_rounds + a.as_rounds + 2); }
template <int NX, int NU>
static int launch_qp_t(slsqp_handle *h, const QpArgs &a, int max_iter, bool mx) {
    // the LDS of a QP wave also holds two n-vectors of the phase logic between the sweeps (phase_update, fused look)
    const size_t lds = std::max(mx ? QpLdsMx<NX, NU>::BYTES : sizeof(double) * (size_t)qp_lds_doubles<NX, NU>(h->d.N), sizeof(double) * (size_t)(2 * h->n + 8));
    const dim3 grid(h->B), blk(64);
    const int max_ticks = qp_max_ticks(a, max_iter);
    // one launch per QP solve: every wave runs its instance to completion (k_qp_solve)
    const bool timed = h->time_kernels && h->n_kev + 2 <= (int)h->kev.size();
    if (timed) hipEventRecord(h->kev[h->n_kev], h->st);
    if (mx) hipLaunchKernelGGL((k_qp_solve<NX, NU, true>), grid, blk, lds, h->st, a, max_ticks);
    else hipLaunchKernelGGL((k_qp_solve<NX, NU, false>), grid, blk, lds, h->st, a, max_ticks);
    if (timed) { hipEventRecord(h->kev[h->n_kev + 1], h->st); h->n_kev += 2; }
    HIPCHK(hipGetLastError());
    return 0;
}

// kernel events recorded by launches that did not wait for their kernels (k_qp_solve): read once the stream has been synchronised
static void harvest_kernel_events(slsqp_handle *h) {
    for (int i = 0; i + 1 < h->n_kev; i += 2) { float ms = 0; if (hipEventElapsedTime(&ms, h->kev[i], h->kev[i + 1]) == hipSuccess) { h->t_fwd += ms; h->n_fwd++; } }
    h->n_kev = 0;
}

// instances of `run` whose mixed-precision solve did not end on the KKT certificate are solved again in fp64
__global__ void k_mark_retry(int B, const int *run, const int *status, int *retry, int *count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int r = (!run || run[b]) && status[b] != 0 && status[b] != 2 && status[b] != 5;      // (2, 5: infeasible for either arithmetic)
    retry[b] = r;
    if (r) atomicAdd(count, 1);
}

static QpArgs make_qp_args(slsqp_handle *h, const int *run, const slsqp_opts *o, int warm, const double *prox = nullptr, int stat_slot = 0, int snap_take = 0, int snap_use = 0,
                           int warm_shift = 0) {
    QpArgs a;
    a.qpstat = h->qpstat; a.stat_slot = stat_slot; a.diag = h->qp_diag;
    static const double snap_mu = getenv("SLSQP_SNAP_MU") ? atof(getenv("SLSQP_SNAP_MU")) : 1e-3;
    a.snap_take = snap_take; a.snap_use = snap_use && o->ipm_restart; a.snap_mu = snap_mu; a.call_id = h->call_id; a.call_ids = h->cl_round ? h->call_ids : nullptr; a.as_first = o->as_first; a.as_rounds = o->as_rounds; a.as_max_viol = o->as_max_viol; a.as_warm_max_set = o->as_warm_max_set; a.as_warm_last = o->as_warm_last;
    { static const double pe = getenv("SLSQP_PINF_EPS") ? atof(getenv("SLSQP_PINF_EPS")) : 1e-4; a.pinf_eps = pe; }
    { static const int ws = getenv("SLSQP_WARM_SHIFT") ? atoi(getenv("SLSQP_WARM_SHIFT")) : 1; a.warm_shift = ws ? warm_shift : 0; a.shift_stepno = nullptr; }
    a.prox = prox; a.prox_stride = 12; a.inst_launches = h->inst_launches;
    a.B = h->B; a.N = h->d.N; a.A = h->A; a.Bm = h->Bm; a.q = h->q; a.ubg = h->ubg; a.lbg = h->lbg; a.x0val = h->x0val; a.run = run;
    a.cst = costs_of(h); a.Linv = h->Linv; a.ws = h->ws; a.primal = h->primal; a.dual = h->dual; a.cost = h->cost; a.pin_dual = h->pin_dual; a.kkt = h->kkt;
    a.status = h->status; a.iters = h->iters; a.max_iter = o->qp_max_iter; a.eps = o->qp_eps;
    a.state = h->qpstate; a.warm = warm; a.warm_rounds = o->warm_rounds;
    a.init_s = getenv("SLSQP_INIT_S") ? atof(getenv("SLSQP_INIT_S")) : 1.0; a.init_lam = getenv("SLSQP_INIT_LAM") ? atof(getenv("SLSQP_INIT_LAM")) : 0.0;
    const bool mx = o->precision == 1;
    a.n_refine = mx ? 3 : 1;   // fp64: one refinement solve; its forward sweep measures the dynamics residual of the first solve (certificate)
    if (getenv("SLSQP_NREFINE")) a.n_refine = atoi(getenv("SLSQP_NREFINE"));
    // tolerance (relative to max(1, |q|inf)) of the look at an un-refined active-set solve.  fp64: the un-refined solve is good to ~1e-11, so the look can
    // use nearly the certificate's own 1e-9 -- with 1e-6 (rounds 1 and 2) violations and wrong-sign multipliers between 1e-9 and 1e-6 went unnoticed until
    // the refined solve's certificate check, and each of those late corrections cost a refinement solve and a fresh factorisation: 9.2 / 11.0 block solves per
    // QP of the rocket loop against 5.9 / 8.0 with 1e-8 (same certified fractions; the numpy prototype of the iteration needs 3.95 rounds, the GPU now 3.8 / 4.7)
    a.early_ctol = mx ? 1e-2 : 1e-8;
    if (!mx && getenv("SLSQP_EARLY_CTOL")) a.early_ctol = atof(getenv("SLSQP_EARLY_CTOL"));      // (experiments)
    return a;
}

static int launch_qp(slsqp_handle *h, const int *run, const slsqp_opts *o, int warm, const double *prox = nullptr, int stat_slot = 0, int snap_take = 0, int snap_use = 0,
                     int warm_shift = 0) {
    const QpArgs a = make_qp_args(h, run, o, warm, prox, stat_slot, snap_take, snap_use, warm_shift);
    const bool mx = o->precision == 1;
    h->time_kernels = o->time_kernels != 0;
    auto go = [&](const QpArgs &q, bool m) {
#define X(NX_, NU_) if (h->d.nx == NX_ && h->d.nu == NU_) return launch_qp_t<NX_, NU_>(h, q, o->qp_max_iter, m);
        SLSQP_DIM_LIST
#undef X
        return -1;
    };
    if (go(a, mx)) return -1;
    h->mx_retry = 0;
    if (mx) {
        HIPCHK(hipMemsetAsync(h->counter + 3, 0, sizeof(int), h->st));
        hipLaunchKernelGGL(k_mark_retry, dim3((h->B + 255) / 256), dim3(256), 0, h->st, h->B, run, h->status, h->retry, h->counter + 3);
        int nretry = 0;
        HIPCHK(hipMemcpyAsync(&nretry, h->counter + 3, sizeof(int), hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        h->mx_retry = nretry; h->mx_retry_total += nretry;
        if (nretry > 0) {
            QpArgs r = a;
            r.run = h->retry; r.warm = 0; r.n_refine = 1; r.early_ctol = 1e-6; r.snap_take = 0; r.snap_use = 0;
            if (go(r, false)) return -1;
        }
    }
    return 0;
}

template <int NX, int NU>
static int launch_sweep_t(slsqp_handle *h, const SweepArgs &a) {
    const size_t lds = sizeof(double) * sweep_lds_doubles<NX, NU>();
    hipLaunchKernelGGL((k_sweep<NX, NU>), dim3(h->B * (h->d.N + 1)), dim3(64), lds, h->st, a);
    HIPCHK(hipGetLastError());
    return 0;
}
template <int NX, int NU>
static int launch_sweep_shared_t(slsqp_handle *h, const SweepArgs &a) {
    SweepSharedArgs aa{a, h->Kc, h->Aclc, h->stale};
    const size_t lds_ric = sizeof(double) * sweep_lds_doubles<NX, NU>(), lds_prop = sizeof(double) * sweep_prop_lds_doubles<NX, NU>();
    hipLaunchKernelGGL((k_sweep_ric1<NX, NU>), dim3(h->B), dim3(64), lds_ric, h->st, aa);
    hipLaunchKernelGGL((k_sweep_prop<NX, NU>), dim3(h->B * ((h->d.N + 2) / 2)), dim3(64), lds_prop, h->st, aa);      // one wave per pair of disturbance columns
    HIPCHK(hipGetLastError());
    return 0;
}
static bool sweep_shared_allowed() { static const bool v = getenv("SLSQP_SWEEP_SHARED") ? atoi(getenv("SLSQP_SWEEP_SHARED")) != 0 : true; return v; }
// shared_cols: every column's eta is the same (first fast-SLS iteration after initialize_backoff): one Riccati recursion per instance
static int launch_sweep(slsqp_handle *h, const int *run, const double *eta, const double *eta_f, double eps, bool shared_cols = false) {
    SweepArgs a;
    a.B = h->B; a.N = h->d.N; a.NW = h->d.nw; a.A = h->A; a.Bm = h->Bm; a.E = h->E; a.E_per_instance = 0; a.eta = eta; a.eta_f = eta_f;
    a.run = run; a.cst = costs_of(h); a.K = h->K; a.beta = h->beta; a.beta_f = h->beta_f; a.ct_part = h->ct_part; a.eps = eps;
    if (h->general_G) {
        SweepGenArgs ga{a, h->Gd, h->Gfd, h->d.ni, h->d.ni_f};
        const dim3 grid(h->B * (h->d.N + 1)), blk(64);
#define X(NX_, NU_) if (h->d.nx == NX_ && h->d.nu == NU_) { const size_t lds = sizeof(double) * sweep_gen_lds_doubles<NX_, NU_>(); hipLaunchKernelGGL((k_sweep_gen<NX_, NU_>), grid, blk, lds, h->st, ga); }
        SLSQP_DIM_LIST
#undef X
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (shared_cols && sweep_shared_allowed()) {
#define X(NX_, NU_) if (h->d.nx == NX_ && h->d.nu == NU_) return launch_sweep_shared_t<NX_, NU_>(h, a);
        SLSQP_DIM_LIST
#undef X
        return -1;
    }
#define X(NX_, NU_) if (h->d.nx == NX_ && h->d.nu == NU_) return launch_sweep_t<NX_, NU_>(h, a);
    SLSQP_DIM_LIST
#undef X
    return -1;
}

// arguments of one RTI chain (k_rti_chain / k_cl_loop): QP #1, what follows it, the sweep, the tightening, QP #2
static int qp_max_ticks(const QpArgs &a, int max_iter);
static ChainArgs make_chain_args(slsqp_handle *h, const slsqp_opts &o, const int *active, int wshift) {
    const slsqp_dims &d = h->d;
    const int B = h->B;
    ChainArgs c;
    c.q1 = make_qp_args(h, h->alive, &o, o.warm_start ? 1 : 0, nullptr, 0, 1, 0, wshift);
    c.q2 = make_qp_args(h, h->alive, &o, 1, nullptr, 1, 1, 1, wshift);
    EtaArgs ea{B, d.N, d.nx, d.ni, d.ni_f, h->dual, h->beta, h->beta_f, h->alive, h->eta, h->eta_f, o.eps_backoff, h->stale, 1};
    ConvArgs ca{B, h->n, h->primal, h->prev_primal, h->has_prev, h->alive, h->conv, o.conv_tol};
    c.aq = AfterQpArgs{B, 1, 1, h->status, active, h->alive, h->infeas, h->mask, h->success, h->itnum, h->counter, h->stale, h->conv, ea, ca, h->beta, h->beta_f};
    SweepArgs sa;
    sa.B = h->B; sa.N = d.N; sa.NW = d.nw; sa.A = h->A; sa.Bm = h->Bm; sa.E = h->E; sa.E_per_instance = 0; sa.eta = h->eta; sa.eta_f = h->eta_f;
    sa.run = h->mask; sa.cst = costs_of(h); sa.K = h->K; sa.beta = h->beta; sa.beta_f = h->beta_f; sa.ct_part = h->ct_part; sa.eps = o.eps_backoff;
    c.sw = SweepSharedArgs{sa, h->Kc, h->Aclc, h->stale};
    c.ta = TightenArgs{B, d.N, d.nx, d.nu, d.ni, d.ni_f, h->beta, h->beta_f, h->g, h->gf_raw, h->c, h->mask, h->backoff, h->backoff_f, h->backoff_x, h->backoff_u, h->ubg, 1, h->ct_part, h->cost_tube};
    c.active = active; c.success = h->success; c.infeas = h->infeas; c.times = h->chain_times;
    c.max_ticks = qp_max_ticks(c.q1, o.qp_max_iter);
    c.lag = nullptr; c.runm = nullptr; c.done = nullptr; c.t0word = nullptr; c.budget = 0; c.qplog = nullptr; c.stepno = nullptr; c.log_steps = 0; c.fin_count = nullptr; c.cut_count = 0xFFFFFFFFu;
    return c;
}

// the fused RTI chain (k_rti_chain): one launch for QP #1 -> eta -> shared Riccati + propagation -> tightened bounds -> QP #2
template <int NX, int NU>
static int launch_chain_t(slsqp_handle *h, ChainArgs &c) {
    const size_t lds = sizeof(double) * std::max({(size_t)qp_lds_doubles<NX, NU>(h->d.N), (size_t)(2 * h->n + 8), (size_t)sweep_lds_doubles<NX, NU>(), (size_t)sweep_prop_lds_doubles<NX, NU>()});
    const bool timed = h->time_kernels && h->n_kev + 2 <= (int)h->kev.size();
    if (timed) hipEventRecord(h->kev[h->n_kev], h->st);
    hipLaunchKernelGGL((k_rti_chain<NX, NU>), dim3(h->B), dim3(64), lds, h->st, c);
    if (timed) { hipEventRecord(h->kev[h->n_kev + 1], h->st); h->n_kev += 2; }
    HIPCHK(hipGetLastError());
    return 0;
}
static bool chain_allowed() { static const bool v = getenv("SLSQP_FUSE_RTI") ? atoi(getenv("SLSQP_FUSE_RTI")) != 0 : true; return v; }

static float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; hipEventElapsedTime(&ms, a, b); return ms; }

// `active` (device, B ints or NULL = all): instances that take part in this call; the others keep every result array untouched.
// ---- timeline of a solve / closed-loop step: event pairs with a role, summed once the stream has been synchronised ----
static int tl_begin(slsqp_handle *h, int role) {
    if (h->tl_n >= (int)h->tl_role.size()) return -1;          // full: this interval goes untimed (flush points keep that from happening)
    const int i = h->tl_n++;
    h->tl_role[i] = role;
    hipEventRecord(h->tl[2 * i], h->st);
    return i;
}
static void tl_end(slsqp_handle *h, int i) { if (i >= 0) hipEventRecord(h->tl[2 * i + 1], h->st); }
// only right after a stream synchronisation: add the recorded intervals to the accumulators, recycle the events
static void tl_flush(slsqp_handle *h) {
    for (int i = 0; i < h->tl_n; i++) { float ms = 0; if (hipEventElapsedTime(&ms, h->tl[2 * i], h->tl[2 * i + 1]) == hipSuccess) h->tl_acc[h->tl_role[i]] += ms; }
    h->tl_n = 0;
    harvest_kernel_events(h);
}
static void tl_take(slsqp_handle *h) {     // accumulators -> the handle's timing fields (what slsqp_last_timing reports)
    tl_flush(h);
    h->t_qp = h->tl_acc[0]; h->t_sweep = h->tl_acc[1]; h->t_total = h->tl_acc[2]; h->t_jac = h->tl_acc[3];
    if (h->tl_acc[4] > 0.0) {
        // fused RTI chain launches: no batch-wide "QP time" / "sweep time" exists any more.  Reported: the sweep part as instance 0 spent it inside
        // the kernel (wall clock, 100 MHz; what the reference's single-instance t_backward_ms measures), the rest of the launches' duration as QP time
        double sw = 0.0;
        if (h->chain_times_host) sw = std::min(h->tl_acc[4], (double)h->chain_times_host[1] * 1e-5);
        h->t_sweep += sw; h->t_qp += h->tl_acc[4] - sw;
    }
    h->tl_acc[0] = h->tl_acc[1] = h->tl_acc[2] = h->tl_acc[3] = h->tl_acc[4] = 0;
}

// no_sync: return with the launches queued (the caller synchronises the stream later and then calls tl_take)
static int solve_impl(slsqp_handle *h, const double *x0, int loc, const slsqp_opts *opts, const int *active, bool no_sync = false) {
    hipSetDevice(h->dev);
    if (h->general_G) return fail("general G: only the sweep-level boundary (slsqp_sweep) is available; the QP solver needs box constraints G = [I;-I]");
    if (!h->have_costs || !h->have_cons || !h->have_dyn) return fail("set_costs, set_constraints and update_dynamics must be called first");
    slsqp_opts o;
    if (opts) o = *opts; else slsqp_default_opts(&o);
    const slsqp_dims &d = h->d;
    const int B = h->B, gb = (B + 255) / 256;
    // x_0 is pinned to -x0 (qp_jit.py:376-379)
    if (loc == SLSQP_HOST) {
        std::vector<double> neg((size_t)B * d.nx);
        for (size_t i = 0; i < neg.size(); i++) neg[i] = -x0[i];
        HIPCHK(hipMemcpyAsync(h->x0val, neg.data(), neg.size() * sizeof(double), hipMemcpyHostToDevice, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
    } else if (!h->beta_inited) {
        hipLaunchKernelGGL(k_negate, dim3((B * d.nx + 255) / 256), dim3(256), 0, h->st, x0, h->x0val, B * d.nx);
    }
    const bool rti = o.rti_steps > 0;
    const int steps = rti ? o.rti_steps : o.max_sls_iter;
    h->call_id += 1.0;
    // the caller moved the horizon one stage on since the last solve (slsqp_cl_step's reset_warm_start): the first QP's warm set moves with it
    const int wshift = h->horizon_shifted; h->horizon_shifted = 0;
    const int tl_tot = tl_begin(h, 2);
    if (h->beta_inited) {     // every solve but the handle's first: the whole preamble in one launch
        SolveBeginArgs sb{B, d.nx, loc == SLSQP_HOST ? nullptr : x0, h->x0val, active, h->alive, h->infeas, h->success, h->pending_reset, h->itnum, h->stale,
                          h->eta, h->eta_f, (size_t)d.N * d.N * d.ni, (size_t)(d.N + 1) * d.ni_f,
                          InitBackoffArgs{B, d.N, d.nx, d.nu, o.eps_backoff, active, h->beta, h->beta_f, h->backoff, h->backoff_f, h->backoff_x, h->backoff_u, 0}, h->cl_skip_begin};
        hipLaunchKernelGGL(k_solve_begin, dim3(B), dim3(256), 0, h->st, sb);
    } else {
    if (active) hipLaunchKernelGGL(k_copy_int, dim3(gb), dim3(256), 0, h->st, active, h->alive, B);
    else hipLaunchKernelGGL(k_fill_int, dim3(gb), dim3(256), 0, h->st, h->alive, 1, B);
    hipLaunchKernelGGL(k_apply_pending_reset, dim3(B), dim3(256), 0, h->st, B, h->pending_reset, active, h->itnum, h->eta, (size_t)d.N * d.N * d.ni, h->eta_f,
                       (size_t)(d.N + 1) * d.ni_f, h->stale);
    hipLaunchKernelGGL(k_fill_int, dim3(gb), dim3(256), 0, h->st, h->infeas, 0, B);
    hipLaunchKernelGGL(k_fill_int, dim3(gb), dim3(256), 0, h->st, h->success, 0, B);
    {   // initialize_backoff at the top of every solve (fast_SLS_jit.py:281,299)
        // the first solve of a handle fills beta for every instance (also the ones outside `active`: later calls rely on it)
        InitBackoffArgs ia{B, d.N, d.nx, d.nu, o.eps_backoff, h->beta_inited ? active : nullptr, h->beta, h->beta_f, h->backoff, h->backoff_f, h->backoff_x, h->backoff_u, h->beta_inited ? 0 : 1};
        if (!h->beta_inited && active) {   // ... but their back-off arrays must stay untouched: two launches then
            InitBackoffArgs ib = ia; ib.run = nullptr;
            InitBackoffArgs ic = ia; ic.run = active; ic.fill_beta = 0;
            hipLaunchKernelGGL(k_fill_doubles, dim3(1024), dim3(256), 0, h->st, h->beta, o.eps_backoff, (size_t)B * d.N * d.N * d.ni);
            hipLaunchKernelGGL(k_fill_doubles, dim3(1024), dim3(256), 0, h->st, h->beta_f, o.eps_backoff, (size_t)B * (d.N + 1) * d.ni_f);
            (void)ib;
            hipLaunchKernelGGL(k_init_backoff, dim3(B), dim3(256), 0, h->st, ic);
        } else hipLaunchKernelGGL(k_init_backoff, dim3(B), dim3(256), 0, h->st, ia);
        h->beta_inited = true;
    }
    }
    // (small batches keep the separate launches: there the SLS propagation of an instance runs as N + 1 waves side by side, which is what a B = 1 caller
    // waits for -- 1.1 against 1.6 ms per rocket RTI step -- while from ~150 instances on the columns of one instance would only compete with other
    // instances' for the same SIMDs.  fuse_rti = 2 forces the chain, slsqp_cl_run always uses it.)
    const bool fuse_here = o.fuse_rti == 2 || h->cl_round || (o.fuse_rti == 1 && (long)B * (d.N + 1) >= 3072);
    if (rti && steps == 1 && o.precision == 0 && fuse_here && chain_allowed() && sweep_shared_allowed() && !h->general_G) {
        // fast_SLS.solve with rti_steps = 1 (the rocket script's setting) as ONE launch: every wave takes its instance through the whole chain
        h->time_kernels = o.time_kernels != 0;
        ChainArgs c = make_chain_args(h, o, active, wshift);
        if (h->cl_round) {      // a round of slsqp_cl_run: suspended solves resume, unfinished ones suspend at the deadline
            c.lag = h->cl_lag; c.runm = h->cl_runm; c.done = h->cl_done; c.t0word = h->t0word; c.budget = h->cl_budget;
            c.qplog = h->qplog; c.stepno = h->cl_stepno; c.log_steps = h->qplog_steps;
            c.fin_count = (unsigned *)(h->t0word + 1); c.cut_count = h->cl_cut_count;
            HIPCHK(hipMemsetAsync(h->t0word, 0, 2 * sizeof(unsigned long long), h->st));
        }
        const int tl_c = tl_begin(h, 4);
        int rc = -1;
#define X(NX_, NU_) if (d.nx == NX_ && d.nu == NU_) rc = launch_chain_t<NX_, NU_>(h, c);
        SLSQP_DIM_LIST
#undef X
        if (rc) return -1;
        tl_end(h, tl_c);
        if (h->chain_times_host) HIPCHK(hipMemcpyAsync(h->chain_times_host, h->chain_times, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->st));
        tl_end(h, tl_tot);
        if (no_sync) return 0;
        HIPCHK(hipStreamSynchronize(h->st));
        tl_take(h);
        return 0;
    }
    for (int i = 0; i < steps; i++) {
        const int tl_q = tl_begin(h, 0);
        if (launch_qp(h, h->alive, &o, (i > 0 || o.warm_start) ? 1 : 0, nullptr, 0, 1, i > 0 ? 1 : 0, i == 0 ? wshift : 0)) return -1;
        tl_end(h, tl_q);
        // flags after the QP, evaluate_dual_eta, check_convergence_socp, masks and (first iteration) the beta repair: one launch
        EtaArgs ea{B, d.N, d.nx, d.ni, d.ni_f, h->dual, h->beta, h->beta_f, h->alive, h->eta, h->eta_f, o.eps_backoff, h->stale, (i == 0 && sweep_shared_allowed()) ? 1 : 0};
        ConvArgs ca{B, h->n, h->primal, h->prev_primal, h->has_prev, h->alive, h->conv, o.conv_tol};
        if (!rti) HIPCHK(hipMemsetAsync(h->counter, 0, sizeof(int), h->st));      // (the count of instances that go on is only read in converge mode)
        AfterQpArgs aq{B, rti ? 1 : 0, i == 0 ? 1 : 0, h->status, active, h->alive, h->infeas, h->mask, h->success, h->itnum, h->counter, h->stale, h->conv, ea, ca, h->beta, h->beta_f};
        hipLaunchKernelGGL(k_after_qp, dim3(B), dim3(256), 0, h->st, aq);
        const int tl_s = tl_begin(h, 1);
        if (launch_sweep(h, h->mask, h->eta, h->eta_f, o.eps_backoff, /* beta == eps for every column right after initialize_backoff */ i == 0)) return -1;
        tl_end(h, tl_s);
        TightenArgs ta{B, d.N, d.nx, d.nu, d.ni, d.ni_f, h->beta, h->beta_f, h->g, h->gf_raw, h->c, h->mask, h->backoff, h->backoff_f, h->backoff_x, h->backoff_u, h->ubg, 1, h->ct_part, h->cost_tube};
        hipLaunchKernelGGL(k_tighten, dim3(B), dim3(128), 0, h->st, ta);
        if (rti) continue;      // RTI mode (every closed-loop script): nothing to decide on the host, the stream runs on
        int nmask = 0;
        HIPCHK(hipMemcpyAsync(&nmask, h->counter, sizeof(int), hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        if (nmask == 0) break;   // every instance converged or failed
    }
    // final QP: RTI always (fast_SLS_jit.py:293); converge mode only for instances that hit MAX_ITER (:311)
    const int tl_q2 = tl_begin(h, 0);
    if (launch_qp(h, h->alive, &o, 1, nullptr, 1, 1, 1, wshift)) return -1;
    tl_end(h, tl_q2);
    hipLaunchKernelGGL(k_finish, dim3(gb), dim3(256), 0, h->st, B, rti ? 1 : 0, h->alive, h->infeas, h->success, active, h->pending_reset);
    tl_end(h, tl_tot);
    if (no_sync) return 0;
    HIPCHK(hipStreamSynchronize(h->st));
    tl_take(h);
    return 0;
}

extern "C" int slsqp_solve(slsqp_handle *h, const double *x0, int loc, const slsqp_opts *opts) { return solve_impl(h, x0, loc, opts, nullptr); }

extern "C" int slsqp_last_timing(slsqp_handle *h, double *ms5, int len) {
    if (!ms5 || len < SLSQP_TIMING_LEN) return fail("slsqp_last_timing: the buffer must hold SLSQP_TIMING_LEN (5) doubles");
    if (h->tl_n > 0) { hipSetDevice(h->dev); hipStreamSynchronize(h->st); tl_take(h); }
    ms5[0] = h->t_total; ms5[1] = h->t_qp; ms5[2] = h->t_sweep; ms5[3] = h->t_total - h->t_qp - h->t_sweep; ms5[4] = h->t_jac;
    return 0;
}
extern "C" int slsqp_kernel_timing(slsqp_handle *h, double *out8, int len) {
    if (!out8 || len < SLSQP_KERNEL_TIMING_LEN) return fail("slsqp_kernel_timing: the buffer must hold SLSQP_KERNEL_TIMING_LEN (8) doubles");
    unsigned long long il[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipSetDevice(h->dev);
    hipMemcpyAsync(il, h->inst_launches, sizeof(il), hipMemcpyDeviceToHost, h->st);
    hipMemsetAsync(h->inst_launches, 0, sizeof(il), h->st);
    hipStreamSynchronize(h->st);
    if (h->tl_n > 0) tl_take(h); else harvest_kernel_events(h);
    out8[0] = h->t_fwd; out8[1] = (double)h->n_fwd; out8[2] = (double)h->mx_retry_total; out8[3] = (double)il[0]; out8[4] = (double)il[1];
    out8[5] = (double)il[2]; out8[6] = (double)il[3]; out8[7] = (double)il[4];
    h->t_fwd = 0; h->n_fwd = 0; h->mx_retry_total = 0;
    return 0;
}

// bytes per instance of a named result (what slsqp_get copies for each instance), or -1 for an unknown name
extern "C" long long slsqp_result_bytes(slsqp_handle *h, const char *name) {
    auto it = h->named.find(name);
    return it == h->named.end() ? -1LL : (long long)it->second.second;
}

extern "C" int slsqp_get(slsqp_handle *h, const char *name, void *out, int loc) {
    hipSetDevice(h->dev);
    auto it = h->named.find(name);
    if (it == h->named.end()) return fail(std::string("unknown result name: ") + name);
    const size_t bytes = it->second.second * (size_t)h->B;
    {   // arrays that slsqp_reset only marked stale
        const slsqp_dims &d = h->d;
        if (!strcmp(name, "eta") || !strcmp(name, "eta_f")) {
            hipLaunchKernelGGL(k_zero_stale, dim3(h->B), dim3(256), 0, h->st, h->stale, 1, h->eta, (size_t)d.N * d.N * d.ni, h->eta_f, (size_t)(d.N + 1) * d.ni_f);
            hipLaunchKernelGGL(k_eta_broadcast, dim3(h->B), dim3(256), 0, h->st, d.N, d.ni, d.ni_f, h->stale, h->eta, h->eta_f);
        }
        else if (!strcmp(name, "K")) {
            hipLaunchKernelGGL(k_zero_stale, dim3(h->B), dim3(256), 0, h->st, h->stale, 2, h->K, (size_t)d.N * (d.N + 1) * d.nu * d.nx, (double *)nullptr, (size_t)0);
            hipLaunchKernelGGL(k_K_broadcast, dim3(h->B), dim3(256), 0, h->st, d.N, d.nu * d.nx, h->stale, h->Kc, h->K);
        }
    }
    HIPCHK(hipMemcpyAsync(out, it->second.first, bytes, loc == SLSQP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    return 0;
}

// Overwrite a named array (the reference's callers poke the QP's bounds between solves: QP.update_ubg, qp_jit.py:578-586, used by
// SCP_SLS_jit.py:86-99).  Writable: ubg, lbg, q, nominal_x, nominal_u, x_meas.
extern "C" int slsqp_set(slsqp_handle *h, const char *name, const void *src, int loc) {
    hipSetDevice(h->dev);
    static const char *ok[] = {"ubg", "lbg", "q", "nominal_x", "nominal_u", "x_meas"};
    bool allowed = false;
    for (const char *k : ok) allowed |= (strcmp(k, name) == 0);
    auto it = h->named.find(name);
    if (!allowed || it == h->named.end()) return fail(std::string("slsqp_set: not a writable array: ") + name);
    const size_t bytes = it->second.second * (size_t)h->B;
    HIPCHK(hipMemcpyAsync(it->second.first, src, bytes, loc == SLSQP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    return 0;
}

extern "C" int slsqp_reset(slsqp_handle *h) {
    hipSetDevice(h->dev);
    // reset_solver_to_zeros (fast_SLS_jit.py:424-442): eta/eta_f/iteration_number to zero, bounds and linear cost dropped.
    // `_prev_primal_vec` is NOT cleared by the reference (quirk q5) and is not cleared here.
    const slsqp_dims &d = h->d;
    const size_t B = h->B;
    // eta, eta_f and K (1.5 GB at rocket B = 4096) are not cleared here: every entry the device reads is rewritten first (evaluate_dual_eta before the
    // sweep, the sweep before anything reads K), so they are only marked stale and zeroed on demand when slsqp_get asks for them
    hipLaunchKernelGGL(k_reset_stale, dim3((h->B + 255) / 256), dim3(256), 0, h->st, h->stale, h->B);
    HIPCHK(hipMemsetAsync(h->itnum, 0, sizeof(int) * B, h->st));
    HIPCHK(hipMemsetAsync(h->pending_reset, 0, sizeof(int) * B, h->st));
    HIPCHK(hipMemsetAsync(h->q, 0, sizeof(double) * B * h->n, h->st));
    (void)d;
    h->have_dyn = false;
    return 0;
}


// ---- linearisation step in front of the path (SCP_SLS.update_jacobian) ----------------------------------------------
extern "C" int slsqp_set_model(slsqp_handle *h, int model_id, const double *g_raw) {
    hipSetDevice(h->dev);
    if (h->general_G) return fail("general G: the plants of slsqp_set_model have box constraints");
    const int want_nx = model_id == 0 ? 4 : (model_id == 1 ? 13 : (model_id == 2 ? 17 : -1));
    if (want_nx != h->d.nx || h->d.nw != h->d.nx) return fail("model id does not match the handle's dimensions (0 pendulum, 1 quadrotor, 2 rocket; nw = nx)");
    HIPCHK(hipMemcpy(h->g_raw, g_raw, sizeof(double) * h->d.ni, hipMemcpyHostToDevice));
    h->model_id = model_id;
    return 0;
}

extern "C" int slsqp_set_E(slsqp_handle *h, const double *E, int loc) {
    hipSetDevice(h->dev);
    return put_E(h, E, loc);
}

static int linearize_impl(slsqp_handle *h, const double *X, const double *U, int loc, const int *run) {
    hipSetDevice(h->dev);
    if (h->model_id < 0 || !h->have_costs || !h->have_cons) return fail("set_model, set_costs and set_constraints must be called first");
    const slsqp_dims &d = h->d;
    const size_t B = h->B;
    const double *dX = X, *dU = U;
    if (loc == SLSQP_HOST) {
        const size_t nX = B * (d.N + 1) * d.nx, nU = B * d.N * d.nu;
        double *tmp = stage_buf(h, sizeof(double) * (nX + nU));
        if (!tmp) return -1;
        HIPCHK(hipMemcpyAsync(tmp, X, sizeof(double) * nX, hipMemcpyHostToDevice, h->st));
        HIPCHK(hipMemcpyAsync(tmp + nX, U, sizeof(double) * nU, hipMemcpyHostToDevice, h->st));
        dX = tmp; dU = tmp + nX;
    }
    LinArgs a{h->B, d.N, dX, dU, h->g_raw, h->gf_raw, costs_of(h), h->A, h->Bm, h->c, h->g, h->gN, h->q, run, h->lin_stage, h->lin_tape};
    const int grid = 2048, blk = 128;
    const int gval = (int)((B * d.N + blk - 1) / blk);
    if (h->model_id == 0) {
        hipLaunchKernelGGL((k_lin_val<0>), dim3(gval), dim3(blk), 0, h->st, a); hipLaunchKernelGGL((k_lin_tan<0>), dim3(grid), dim3(blk), 0, h->st, a);
        hipLaunchKernelGGL((k_lin_vec<4, 1>), dim3(grid), dim3(256), 0, h->st, a);
    } else if (h->model_id == 1) {
        hipLaunchKernelGGL((k_lin_val<1>), dim3(gval), dim3(blk), 0, h->st, a); hipLaunchKernelGGL((k_lin_tan<1>), dim3(grid), dim3(blk), 0, h->st, a);
        hipLaunchKernelGGL((k_lin_vec<13, 4>), dim3(grid), dim3(256), 0, h->st, a);
    } else {
        hipLaunchKernelGGL((k_lin_val<2>), dim3(gval), dim3(blk), 0, h->st, a); hipLaunchKernelGGL((k_lin_tan<2>), dim3(grid), dim3(blk), 0, h->st, a);
        hipLaunchKernelGGL((k_lin_vec<17, 4>), dim3(grid), dim3(256), 0, h->st, a);
    }
    BoundsArgs ba{h->B, d.N, d.nx, d.ni, d.ni_f, h->g, h->gN, h->c, h->ubg, h->lbg, 1e-10, run};
    hipLaunchKernelGGL(k_set_bounds, dim3(1024), dim3(256), 0, h->st, ba);
    HIPCHK(hipGetLastError());
    if (loc == SLSQP_HOST) HIPCHK(hipStreamSynchronize(h->st));   // the caller's buffers may be reused on return; device callers stay asynchronous on the handle's stream
    h->have_dyn = true;
    return 0;
}
extern "C" int slsqp_linearize(slsqp_handle *h, const double *X, const double *U, int loc) { return linearize_impl(h, X, U, loc, nullptr); }


// ---- closed-loop driver around the path (SCP_SLS.solve + reset_warm_start + plant) -------------------------
static ClArgs cl_args(slsqp_handle *h, const double *w) {
    ClArgs a;
    a.B = h->B; a.N = h->d.N; a.NX = h->d.nx; a.NU = h->d.nu; a.Xn = h->Xn; a.Un = h->Un; a.xmeas = h->xmeas; a.primal = h->primal;
    a.success = h->success; a.x0arg = h->x0arg; a.E = h->E; a.w = w; a.u0 = h->u0; a.u_init = h->u_init;
    return a;
}

extern "C" int slsqp_cl_init(slsqp_handle *h, const double *x_meas, const double *X_nom, const double *U_nom, const double *u_init, int loc) {
    hipSetDevice(h->dev);
    if (h->model_id < 0) return fail("slsqp_set_model must be called first");
    const slsqp_dims &d = h->d;
    const size_t B = h->B;
    if (put(h, h->xmeas, x_meas, sizeof(double) * B * d.nx, loc)) return -1;
    if (X_nom && U_nom) {
        if (put(h, h->Xn, X_nom, sizeof(double) * B * (d.N + 1) * d.nx, loc)) return -1;
        if (put(h, h->Un, U_nom, sizeof(double) * B * d.N * d.nu, loc)) return -1;
    } else {
        std::vector<double> ui(d.nu, 0.0);
        if (u_init) for (int i = 0; i < d.nu; i++) ui[i] = u_init[i];
        HIPCHK(hipMemcpy(h->u_init, ui.data(), sizeof(double) * d.nu, hipMemcpyHostToDevice));
        ClArgs a = cl_args(h, nullptr);
        const int gb = (h->B + 63) / 64;
        if (h->model_id == 0) hipLaunchKernelGGL((k_cl_rollout<0>), dim3(gb), dim3(64), 0, h->st, a);
        else if (h->model_id == 1) hipLaunchKernelGGL((k_cl_rollout<1>), dim3(gb), dim3(64), 0, h->st, a);
        else hipLaunchKernelGGL((k_cl_rollout<2>), dim3(gb), dim3(64), 0, h->st, a);
        HIPCHK(hipGetLastError());
    }
    h->cl_steps = 0;
    HIPCHK(hipMemsetAsync(h->has_prev, 0, sizeof(int) * B, h->st));   // a fresh SCP_SLS object has no convergence history
    HIPCHK(hipStreamSynchronize(h->st));
    return slsqp_reset(h);
}

// Nominal-trajectory initialiser: trust-region SCP on the path's QP kernel (see k_nom_eval).  Works on the nominal held by the
// handle (slsqp_cl_init: caller's guess or roll-out) and the measured state given there.
extern "C" int slsqp_nominal_solve(slsqp_handle *h, int max_qp, double tol, double rho, const slsqp_opts *opts) {
    hipSetDevice(h->dev);
    if (h->general_G) return fail("general G: only the sweep-level boundary (slsqp_sweep) is available; the QP solver needs box constraints G = [I;-I]");
    if (h->model_id < 0 || !h->have_costs || !h->have_cons) return fail("set_model, set_costs and set_constraints must be called first");
    slsqp_opts o;
    if (opts) o = *opts; else slsqp_default_opts(&o);
    const slsqp_dims &d = h->d;
    const int B = h->B, gbi = (B + 255) / 256;
    if (max_qp <= 0) max_qp = 120;
    if (!(tol > 0.0)) tol = 1e-7;
    if (!(rho > 0.0)) rho = 1e3;
    int *active = h->scp_active;
    hipLaunchKernelGGL(k_nom_init, dim3(gbi), dim3(256), 0, h->st, B, h->nom_st, active, h->nom_need_lin, h->nom_status, h->nom_iters, 1.0, 0.5);
    ClArgs ca = cl_args(h, nullptr);
    NomArgs na;
    na.B = B; na.N = d.N; na.Xn = h->Xn; na.Un = h->Un; na.xmeas = h->xmeas; na.primal = h->primal; na.qp_status = h->status; na.g_raw = h->g_raw; na.gf_raw = h->gf_raw;
    na.cst = costs_of(h); na.st = h->nom_st; na.active = active; na.need_lin = h->nom_need_lin; na.status = h->nom_status; na.iters = h->nom_iters;
    na.n_active = h->counter + 2; na.rho = rho; na.tol = tol; na.w_max = 1e8;
    auto eval = [&](int mode) {
        na.mode = mode;
        if (h->model_id == 0) hipLaunchKernelGGL((k_nom_eval<0>), dim3(B), dim3(128), 0, h->st, na);
        else if (h->model_id == 1) hipLaunchKernelGGL((k_nom_eval<1>), dim3(B), dim3(128), 0, h->st, na);
        else hipLaunchKernelGGL((k_nom_eval<2>), dim3(B), dim3(128), 0, h->st, na);
    };
    eval(0);
    double tq = 0;
    for (int it = 0; it < max_qp; it++) {
        if (linearize_impl(h, h->Xn, h->Un, SLSQP_DEVICE, h->nom_need_lin)) return -1;     // accepted instances only; the others re-solve
        NomBoundsArgs ba{B, d.N, d.nx, d.ni, d.ni_f, h->g, h->gN, h->c, h->nom_st, active, h->ubg, h->lbg, 1e-10};
        hipLaunchKernelGGL(k_nom_bounds, dim3(1024), dim3(256), 0, h->st, ba);
        hipLaunchKernelGGL(k_nom_x0, dim3((B * d.nx + 255) / 256), dim3(256), 0, h->st, B, d.N, d.nx, h->Xn, h->xmeas, h->nom_st, h->x0val);
        HIPCHK(hipEventRecord(h->ev[6], h->st));
        if (launch_qp(h, active, &o, it > 0 ? 1 : 0, h->nom_st /* S[0] = w: stride 12 */)) return -1;
        HIPCHK(hipEventRecord(h->ev[7], h->st));
        HIPCHK(hipMemsetAsync(h->counter + 2, 0, sizeof(int), h->st));
        eval(1);
        int nact = 0;
        HIPCHK(hipMemcpyAsync(&nact, h->counter + 2, sizeof(int), hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        harvest_kernel_events(h);
        tq += ev_ms(h->ev[6], h->ev[7]);
        if (nact == 0) break;
    }
    h->t_qp = tq; h->t_total = tq; h->t_sweep = 0;
    h->have_dyn = false;   // the bounds in the handle are the initialiser's, not the path's: linearise again before slsqp_solve
    return 0;
}

// Device-side log of the closed loop: after this call every slsqp_cl_step stores what the reference's scripts keep per MPC step
// (expe/main_rocket_robust_closed_loop.py:160-178) into entry cl_steps of (B, max_steps, ...) buffers; slsqp_get names: log_state (S,nx)
// log_u0 (S,nu) log_nominal_x (S,N+1,nx) log_nominal_u (S,N,nu) log_backoff_x (S,N+1,nx) log_backoff_u (S,N,nu) log_success[int32] (S)
// log_scp_iterations[int32] (S).  slsqp_cl_init restarts at entry 0.
extern "C" int slsqp_cl_log(slsqp_handle *h, int max_steps) {
    hipSetDevice(h->dev);
    if (max_steps < 1) return fail("slsqp_cl_log: max_steps must be >= 1");
    HIPCHK(hipStreamSynchronize(h->st));
    h->log_steps = 0;
    free_all(h->log_owned);
    static const char *names[] = {"log_nominal_x", "log_nominal_u", "log_backoff_x", "log_backoff_u", "log_state", "log_u0", "log_success", "log_scp_iterations", "log_primal_infeasibility"};
    for (const char *nm : names) h->named.erase(nm);
    h->lg_x = h->lg_u = h->lg_bx = h->lg_bu = h->lg_state = h->lg_u0 = h->lg_pinf = nullptr; h->lg_succ = h->lg_it = nullptr;
    const slsqp_dims &d = h->d;
    const size_t B = h->B, S = max_steps, nX = (size_t)(d.N + 1) * d.nx, nU = (size_t)d.N * d.nu;
    int rc = 0;
    auto &ow = h->log_owned;
    rc |= dalloc(ow, &h->lg_x, B * S * nX); rc |= dalloc(ow, &h->lg_u, B * S * nU); rc |= dalloc(ow, &h->lg_bx, B * S * nX); rc |= dalloc(ow, &h->lg_bu, B * S * nU);
    rc |= dalloc(ow, &h->lg_state, B * S * d.nx); rc |= dalloc(ow, &h->lg_u0, B * S * d.nu); rc |= dalloc(ow, &h->lg_succ, B * S); rc |= dalloc(ow, &h->lg_it, B * S);
    rc |= dalloc(ow, &h->lg_pinf, B * S);
    if (rc) { free_all(ow); h->lg_x = h->lg_u = h->lg_bx = h->lg_bu = h->lg_state = h->lg_u0 = h->lg_pinf = nullptr; h->lg_succ = h->lg_it = nullptr; return -1; }
    h->log_steps = max_steps;
    auto reg = [&](const char *nm, void *p, size_t bytes) { h->named[nm] = {p, bytes}; };
    reg("log_nominal_x", h->lg_x, sizeof(double) * S * nX); reg("log_nominal_u", h->lg_u, sizeof(double) * S * nU);
    reg("log_backoff_x", h->lg_bx, sizeof(double) * S * nX); reg("log_backoff_u", h->lg_bu, sizeof(double) * S * nU);
    reg("log_state", h->lg_state, sizeof(double) * S * d.nx); reg("log_u0", h->lg_u0, sizeof(double) * S * d.nu);
    reg("log_success", h->lg_succ, sizeof(int) * S); reg("log_scp_iterations", h->lg_it, sizeof(int) * S);
    reg("log_primal_infeasibility", h->lg_pinf, sizeof(double) * S);
    return 0;
}

// One MPC step for the whole batch: [warm-start shift + solver reset] -> rti x (linearise, fast-SLS solve of x_nom0 - x_meas,
// nominal += delta) -> u0 = first nominal input -> plant step x_meas <- ddyn(x_meas,u0) + E w.   w (B,nx) or NULL (no noise).
extern "C" int slsqp_cl_step(slsqp_handle *h, int rti, const double *w, int loc, const slsqp_opts *opts) {
    hipSetDevice(h->dev);
    if (h->model_id < 0) return fail("slsqp_set_model must be called first");
    const slsqp_dims &d = h->d;
    const int gb = (h->B + 63) / 64;
    const double *dw = nullptr;
    if (w) { if (put(h, h->wbuf, w, sizeof(double) * (size_t)h->B * d.nx, loc)) return -1; dw = h->wbuf; }
    ClArgs a = cl_args(h, dw);
    if (h->cl_steps > 0) {
        if (h->model_id == 0) hipLaunchKernelGGL((k_cl_shift_plant<0>), dim3(gb), dim3(64), 0, h->st, a, 1, 0);
        else if (h->model_id == 1) hipLaunchKernelGGL((k_cl_shift_plant<1>), dim3(gb), dim3(64), 0, h->st, a, 1, 0);
        else hipLaunchKernelGGL((k_cl_shift_plant<2>), dim3(gb), dim3(64), 0, h->st, a, 1, 0);
        if (slsqp_reset(h)) return -1;
        h->horizon_shifted = 1;
    }
    slsqp_opts o;
    if (opts) o = *opts; else slsqp_default_opts(&o);
    // SCP_SLS.solve (solver/SCP_SLS_jit.py:65-152): rti > 0 -> exactly rti iterations; rti <= 0 -> until |delta|inf < scp_eps
    // (epsilon_convergence :29) or MAX_ITER_SCP (:47).  Instances leave the loop one by one (failed step / converged); the
    // linearisation, the fast-SLS solve and the nominal update only touch the ones still iterating.
    const bool converge = rti <= 0;
    const int max_it = converge ? o.max_scp_iter : rti;
    const int B = h->B, gbi = (B + 255) / 256;
    hipLaunchKernelGGL(k_fill_int, dim3(gbi), dim3(256), 0, h->st, h->scp_active, 1, B);
    hipLaunchKernelGGL(k_fill_int, dim3(gbi), dim3(256), 0, h->st, h->scp_success, 0, B);
    hipLaunchKernelGGL(k_fill_int, dim3(gbi), dim3(256), 0, h->st, h->scp_iters, 0, B);
    {
        const int tl_j = tl_begin(h, 3);
        if (linearize_impl(h, h->Xn, h->Un, SLSQP_DEVICE, nullptr)) return -1;
        tl_end(h, tl_j);
    }
    for (int ii = 0; ii < max_it; ii++) {
        hipLaunchKernelGGL(k_cl_x0arg, dim3(64), dim3(256), 0, h->st, a);
        if (solve_impl(h, h->x0arg, SLSQP_DEVICE, &o, h->scp_active, /* no_sync */ true)) return -1;
        HIPCHK(hipMemsetAsync(h->counter + 2, 0, sizeof(int), h->st));
        ScpArgs sa{ii, converge ? 1 : 0, o.scp_eps, h->scp_active, h->scp_success, h->scp_iters, h->counter + 2, h->scp_dmax, h->scp_upd};
        hipLaunchKernelGGL(k_cl_scp_update, dim3(B), dim3(64), 0, h->st, a, sa);
        if (h->model_id == 0) hipLaunchKernelGGL((k_cl_infeas<0>), dim3(B), dim3(64), 0, h->st, a, h->scp_upd, h->pinf);
        else if (h->model_id == 1) hipLaunchKernelGGL((k_cl_infeas<1>), dim3(B), dim3(64), 0, h->st, a, h->scp_upd, h->pinf);
        else hipLaunchKernelGGL((k_cl_infeas<2>), dim3(B), dim3(64), 0, h->st, a, h->scp_upd, h->pinf);
        if (ii + 1 == max_it) break;
        if (converge) {     // the host decides whether anyone is still iterating; RTI mode runs its rti iterations without looking (the instances that
                            // failed are masked on the device), so a whole RTI step is one burst of launches on the stream
            int nact = 0;
            HIPCHK(hipMemcpyAsync(&nact, h->counter + 2, sizeof(int), hipMemcpyDeviceToHost, h->st));
            HIPCHK(hipStreamSynchronize(h->st));
            tl_flush(h);
            if (nact == 0) break;
        }
        const int tl_j = tl_begin(h, 3);
        if (linearize_impl(h, h->Xn, h->Un, SLSQP_DEVICE, h->scp_active)) return -1;   // update_jacobian for the next iteration (:138)
        tl_end(h, tl_j);
    }
    if (h->log_steps > 0 && h->cl_steps < h->log_steps) {
        ClLogArgs la{nullptr, nullptr, h->B, d.N, d.nx, d.nu, h->log_steps, h->cl_steps, h->Xn, h->Un, h->backoff_x, h->backoff_u, h->scp_success, h->scp_iters,
                     h->pinf, h->lg_x, h->lg_u, h->lg_bx, h->lg_bu, h->lg_state, h->lg_u0, h->lg_pinf, h->lg_succ, h->lg_it};
        hipLaunchKernelGGL(k_cl_log, dim3(1024), dim3(256), 0, h->st, la);
    }
    if (h->model_id == 0) hipLaunchKernelGGL((k_cl_shift_plant<0>), dim3(gb), dim3(64), 0, h->st, a, 0, 1);
    else if (h->model_id == 1) hipLaunchKernelGGL((k_cl_shift_plant<1>), dim3(gb), dim3(64), 0, h->st, a, 0, 1);
    else hipLaunchKernelGGL((k_cl_shift_plant<2>), dim3(gb), dim3(64), 0, h->st, a, 0, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->st));
    tl_take(h);
    h->cl_steps++;
    return 0;
}

// ---- the whole closed loop with instances advancing independently ------------------------------------------------------------------
__global__ void k_cl_begin_flags(int B, const int *begin, int *scp_success, int *scp_iters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && begin[b]) { scp_success[b] = 0; scp_iters[b] = 0; }
}
// the persistent closed-loop launch (k_cl_loop): as many waves as the GPU holds at the kernel's occupancy (or as there are instances)
template <int MODEL>
static int launch_loop_t(slsqp_handle *h, LoopArgs &L) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    const size_t lds = sizeof(double) * std::max({(size_t)qp_lds_doubles<NX, NU>(h->d.N), (size_t)(2 * h->n + 8), (size_t)sweep_lds_doubles<NX, NU>(), (size_t)sweep_prop_lds_doubles<NX, NU>()});
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, h->dev) != hipSuccess || cu <= 0) cu = 256;
    const int env_waves = getenv("SLSQP_LOOP_WAVES") ? atoi(getenv("SLSQP_LOOP_WAVES")) : 0;      // (tests / experiments: fewer waves than the GPU holds)
    const int slots = env_waves > 0 ? env_waves : cu * 4 * QP_PERSIST_WAVES_PER_SIMD;
    const int grid = std::max(1, std::min(h->B, slots));
    h->cl_loop_waves = grid;
    const bool timed = h->time_kernels && h->n_kev + 2 <= (int)h->kev.size();
    if (timed) hipEventRecord(h->kev[h->n_kev], h->st);
    hipEventRecord(h->ev[8], h->st);
    hipLaunchKernelGGL((k_cl_loop<MODEL>), dim3(grid), dim3(64), lds, h->st, L);
    hipEventRecord(h->ev[9], h->st);
    if (timed) { hipEventRecord(h->kev[h->n_kev + 1], h->st); h->n_kev += 2; }
    HIPCHK(hipGetLastError());
    return 0;
}
static int cl_run_persistent(slsqp_handle *h, int steps, const double *dW, const slsqp_opts &o, int *rounds_out) {
    const slsqp_dims &d = h->d;
    const int B = h->B;
    if (!h->beta_inited) {      // what the handle's first solve does once: beta = eps everywhere (later solves only repair swept instances)
        hipLaunchKernelGGL(k_fill_doubles, dim3(1024), dim3(256), 0, h->st, h->beta, o.eps_backoff, (size_t)B * d.N * d.N * d.ni);
        hipLaunchKernelGGL(k_fill_doubles, dim3(1024), dim3(256), 0, h->st, h->beta_f, o.eps_backoff, (size_t)B * (d.N + 1) * d.ni_f);
        h->beta_inited = true;
    }
    h->time_kernels = o.time_kernels != 0;
    LoopArgs L;
    L.c = make_chain_args(h, o, nullptr, 1);
    L.c.q1.call_ids = L.c.q2.call_ids = h->call_ids;
    L.c.q1.shift_stepno = L.c.q2.shift_stepno = h->cl_stepno;      // the horizon has moved for the instances past their first step
    L.c.qplog = h->qplog; L.c.stepno = h->cl_stepno; L.c.log_steps = h->qplog_steps;
    L.cl = cl_args(h, nullptr);
    L.lin = LinArgs{B, d.N, h->Xn, h->Un, h->g_raw, h->gf_raw, costs_of(h), h->A, h->Bm, h->c, h->g, h->gN, h->q, nullptr, h->lin_stage, h->lin_tape};
    L.ba = BoundsArgs{B, d.N, d.nx, d.ni, d.ni_f, h->g, h->gN, h->c, h->ubg, h->lbg, 1e-10, nullptr};
    L.sb = SolveBeginArgs{B, d.nx, h->x0arg, h->x0val, nullptr, h->alive, h->infeas, h->success, h->pending_reset, h->itnum, h->stale,
                          h->eta, h->eta_f, (size_t)d.N * d.N * d.ni, (size_t)(d.N + 1) * d.ni_f,
                          InitBackoffArgs{B, d.N, d.nx, d.nu, o.eps_backoff, nullptr, h->beta, h->beta_f, h->backoff, h->backoff_f, h->backoff_x, h->backoff_u, 0}, nullptr};
    L.lg = ClLogArgs{h->cl_stepno, nullptr, B, d.N, d.nx, d.nu, h->log_steps, 0, h->Xn, h->Un, h->backoff_x, h->backoff_u, h->scp_success, h->scp_iters,
                     h->pinf, h->lg_x, h->lg_u, h->lg_bx, h->lg_bu, h->lg_state, h->lg_u0, h->lg_pinf, h->lg_succ, h->lg_it};
    L.sa = ScpArgs{0, 0, o.scp_eps, h->scp_active, h->scp_success, h->scp_iters, h->counter + 2, h->scp_dmax, h->scp_upd};
    L.steps = steps; L.n = h->n; L.have_log = h->log_steps > 0 ? 1 : 0;
    L.keep_laggards = getenv("SLSQP_LOOP_KEEP") ? atoi(getenv("SLSQP_LOOP_KEEP")) : 1;
    L.fence = getenv("SLSQP_LOOP_FENCE") ? atoi(getenv("SLSQP_LOOP_FENCE")) : 3;      // (experiments only: 0 drops the hand-over fences)
    L.stepno = h->cl_stepno; L.call_ids = h->call_ids; L.q = h->q; L.stale = h->stale; L.itnum = h->itnum; L.pending = h->pending_reset; L.scp_active = h->scp_active;
    L.W_all = dW; L.pinf = h->pinf;
    L.Q = ClQueue{h->clq_slots, h->clq_cap - 1u, (unsigned *)h->clq_ctl, (unsigned *)h->clq_ctl + 1, h->clq_ctl + 2, h->clq_ctl + 3};
    L.busy = h->cl_busy; L.t_begin = h->cl_tbegin;
    hipLaunchKernelGGL(k_clq_init, dim3(64), dim3(256), 0, h->st, B, L.Q);
    HIPCHK(hipMemsetAsync(h->cl_busy, 0, 16 * sizeof(unsigned long long), h->st));
    HIPCHK(hipMemsetAsync(h->cl_busy + 14, 0xFF, sizeof(unsigned long long), h->st));      // (CL_LOOP_STAMP builds: earliest wave start)
    HIPCHK(hipMemsetAsync(h->counter + 2, 0, sizeof(int), h->st));
    const int tl_tot = tl_begin(h, 2), tl_c = tl_begin(h, 4);
    int rc = -1;
    if (h->model_id == 0) rc = launch_loop_t<0>(h, L);
    else if (h->model_id == 1) rc = launch_loop_t<1>(h, L);
    else rc = launch_loop_t<2>(h, L);
    if (rc) return -1;
    tl_end(h, tl_c); tl_end(h, tl_tot);
    int ctl[4] = {0, 0, 0, 0};
    std::vector<int> sn((size_t)B);
    HIPCHK(hipMemcpyAsync(ctl, h->clq_ctl, sizeof(ctl), hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipMemcpyAsync(sn.data(), h->cl_stepno, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipMemcpyAsync(h->cl_busy_host, h->cl_busy, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->st));
    if (h->chain_times_host) HIPCHK(hipMemcpyAsync(h->chain_times_host, h->chain_times, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    tl_take(h);
    h->cl_loop_ms = ev_ms(h->ev[8], h->ev[9]);
    h->have_dyn = true;
    int unfinished = 0;
    for (int b = 0; b < B; b++) unfinished += sn[b] < steps ? 1 : 0;
    if (ctl[3] != 0 || unfinished != 0)
        return fail("slsqp_cl_run (persistent): the instance queue did not drain (err " + std::to_string(ctl[3]) + ", " + std::to_string(unfinished) + " instances short of their steps)");
    h->cl_steps = steps;
    h->call_id += steps;
    if (rounds_out) *rounds_out = 1;
    return 0;
}
// wave statistics of the last persistent slsqp_cl_run: [0] waves launched, [1] sum over the MPC steps of the time a wave spent on them (ms),
// [2] MPC steps run, [3] duration of the launch (ms; HIP events, opts.time_kernels)
extern "C" int slsqp_cl_run_stats(slsqp_handle *h, double *out, int len) {
    if (!out || len < SLSQP_CL_RUN_STATS_LEN) return fail("slsqp_cl_run_stats: the buffer must hold SLSQP_CL_RUN_STATS_LEN (4) doubles");
    out[0] = (double)h->cl_loop_waves; out[1] = (double)h->cl_busy_host[0] * 1e-5; out[2] = (double)h->cl_busy_host[2]; out[3] = h->cl_loop_ms;
    for (int i = 4; i < 16 && i < len; i++) out[i] = (double)h->cl_busy_host[i] * 1e-5;
    if (len >= 16) { out[14] = (double)(h->cl_busy_host[15] - h->cl_busy_host[14]) * 1e-5; out[15] = (double)h->cl_busy_host[1]; }      // first wave start -> last wave exit; waves that ran      // -DCL_LOOP_STAMP builds: ms spent in the parts of the step around the chain
    return 0;
}

// `steps` MPC steps of every instance (rti = 1 with one fast-SLS step: the rocket script's setting), W (steps, B, nx) or NULL.  Per instance the
// same sequence of operations as `steps` calls of slsqp_cl_step -- same bits -- but no instance waits for another: the loop runs in ROUNDS (one
// burst of launches each); in a round an instance either begins its next MPC step (shift, reset, linearise, chain, nominal update, plant) or
// resumes the QP solve a previous round's deadline suspended; a chain that is not done `budget_ms` after its launch started suspends itself
// between two block solves and the instance simply takes part in the next round where it stopped.  A batch-wide step lasts as long as its slowest
// instance (50-90 block solves against a mean of 20: two thirds of a step's time is spent waiting for a few per cent of the instances); a round
// lasts budget_ms at most, and ends earlier once cut_frac of its participants are done (0 < cut_frac < 1; the stragglers of a round then are cut as soon
// as the bulk has finished, whatever time that took).  budget_ms <= 0: no time limit.  Results: the device-side log (slsqp_cl_log with max_steps >= steps), the final state, and log_qp_stats (steps, 2, 8).
extern "C" int slsqp_cl_run(slsqp_handle *h, int steps, const double *W, int loc, const slsqp_opts *opts, double budget_ms, double cut_frac, int *rounds_out) {
    hipSetDevice(h->dev);
    if (h->model_id < 0) return fail("slsqp_set_model must be called first");
    if (h->cl_steps != 0) return fail("slsqp_cl_run starts a closed loop: call slsqp_cl_init first");
    if (steps < 1) return fail("slsqp_cl_run: steps must be >= 1");
    if (h->log_steps > 0 && h->log_steps < steps) return fail("slsqp_cl_run: the device-side log (slsqp_cl_log) is shorter than the run");
    slsqp_opts o;
    if (opts) o = *opts; else slsqp_default_opts(&o);
    if (!(o.rti_steps == 1 && o.precision == 0 && o.fuse_rti && chain_allowed() && sweep_shared_allowed()) || h->general_G)
        return fail("slsqp_cl_run needs the fused RTI chain: rti_steps = 1, fp64, fuse_rti = 1, box constraints (use slsqp_cl_step otherwise)");
    const slsqp_dims &d = h->d;
    const int B = h->B, gbi = (B + 255) / 256, gb = (B + 63) / 64;
    const double *dW = nullptr;
    if (W) {
        const size_t nW = (size_t)steps * B * d.nx;
        if (loc == SLSQP_HOST) {
            if (nW > h->cl_W_doubles) { if (h->cl_W) hipFree(h->cl_W); h->cl_W = nullptr; h->cl_W_doubles = 0; HIPCHK(hipMalloc((void **)&h->cl_W, nW * sizeof(double) + 64)); h->cl_W_doubles = nW; }
            HIPCHK(hipMemcpyAsync(h->cl_W, W, nW * sizeof(double), hipMemcpyHostToDevice, h->st));
            dW = h->cl_W;
        } else dW = W;
    }
    if (h->qplog_steps < steps) {
        if (h->qplog) hipFree(h->qplog);
        h->qplog = nullptr; h->qplog_steps = 0;
        HIPCHK(hipMalloc((void **)&h->qplog, (size_t)B * steps * 16 * sizeof(int) + 64));
        h->qplog_steps = steps;
    }
    HIPCHK(hipMemsetAsync(h->qplog, 0, (size_t)B * h->qplog_steps * 16 * sizeof(int), h->st));
    h->named["log_qp_stats"] = {h->qplog, sizeof(int) * 16 * (size_t)h->qplog_steps};
    HIPCHK(hipMemsetAsync(h->cl_stepno, 0, sizeof(int) * B, h->st));
    HIPCHK(hipMemsetAsync(h->cl_lag, 0, sizeof(int) * B, h->st));
    hipLaunchKernelGGL(k_fill_doubles, dim3(64), dim3(256), 0, h->st, h->call_ids, h->call_id, (size_t)B);
    if (o.cl_persistent) return cl_run_persistent(h, steps, dW, o, rounds_out);
    h->cl_budget = budget_ms > 0.0 ? (unsigned long long)(std::max(0.05, budget_ms) * 1e5) : (1ULL << 62);
    h->cl_total_steps = steps;
    ClArgs a = cl_args(h, nullptr);
    int rounds = 0;
    const int max_rounds = 40 * steps + 100;
    for (;; rounds++) {
        int unfinished = 0;
        HIPCHK(hipMemsetAsync(h->counter + 1, 0, sizeof(int), h->st));
        hipLaunchKernelGGL(k_cl_round_masks, dim3(gbi), dim3(256), 0, h->st, B, steps, h->cl_stepno, h->cl_lag, h->cl_begin, h->cl_runm, h->cl_skipb, h->cl_done, h->counter + 1);
        HIPCHK(hipMemcpyAsync(&unfinished, h->counter + 1, sizeof(int), hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        tl_flush(h);
        if (unfinished == 0) break;
        h->cl_cut_count = (cut_frac > 0.0 && cut_frac < 1.0) ? (unsigned)std::max(1.0, std::ceil(cut_frac * unfinished)) : 0xFFFFFFFFu;
        if (rounds >= max_rounds) { h->cl_round = false; h->cl_skip_begin = nullptr; return fail("slsqp_cl_run: round limit reached"); }
        // begin: reset_warm_start (shift + solver reset) for the instances past their first step
        if (h->model_id == 0) hipLaunchKernelGGL((k_cl_shift_plant<0>), dim3(gb), dim3(64), 0, h->st, a, 1, 0, h->cl_begin, h->cl_stepno, nullptr);
        else if (h->model_id == 1) hipLaunchKernelGGL((k_cl_shift_plant<1>), dim3(gb), dim3(64), 0, h->st, a, 1, 0, h->cl_begin, h->cl_stepno, nullptr);
        else hipLaunchKernelGGL((k_cl_shift_plant<2>), dim3(gb), dim3(64), 0, h->st, a, 1, 0, h->cl_begin, h->cl_stepno, nullptr);
        hipLaunchKernelGGL(k_cl_reset_masked, dim3(B), dim3(256), 0, h->st, B, h->n, h->cl_begin, h->cl_stepno, h->stale, h->itnum, h->pending_reset, h->q);
        hipLaunchKernelGGL(k_cl_bump_call, dim3(gbi), dim3(256), 0, h->st, B, h->cl_begin, h->call_ids);
        hipLaunchKernelGGL(k_cl_begin_flags, dim3(gbi), dim3(256), 0, h->st, B, h->cl_begin, h->scp_success, h->scp_iters);
        h->have_dyn = false;
        {
            const int tl_j = tl_begin(h, 3);
            if (linearize_impl(h, h->Xn, h->Un, SLSQP_DEVICE, h->cl_begin)) return -1;
            tl_end(h, tl_j);
        }
        hipLaunchKernelGGL(k_cl_x0arg, dim3(64), dim3(256), 0, h->st, a, h->cl_begin);
        h->horizon_shifted = rounds > 0 ? 1 : 0;
        h->cl_round = true; h->cl_skip_begin = h->cl_skipb;
        const int rc = solve_impl(h, h->x0arg, SLSQP_DEVICE, &o, h->cl_runm, /* no_sync */ true);
        h->cl_round = false; h->cl_skip_begin = nullptr;
        if (rc) return -1;
        // end of the step for the instances whose chain is done: nominal += delta, primal_infeasibility, log entry, plant + noise, step counter
        hipLaunchKernelGGL(k_copy_int, dim3(gbi), dim3(256), 0, h->st, h->cl_done, h->scp_active, B);
        HIPCHK(hipMemsetAsync(h->counter + 2, 0, sizeof(int), h->st));
        ScpArgs sa{0, 0, o.scp_eps, h->scp_active, h->scp_success, h->scp_iters, h->counter + 2, h->scp_dmax, h->scp_upd};
        hipLaunchKernelGGL(k_cl_scp_update, dim3(B), dim3(64), 0, h->st, a, sa);
        if (h->model_id == 0) hipLaunchKernelGGL((k_cl_infeas<0>), dim3(B), dim3(64), 0, h->st, a, h->scp_upd, h->pinf);
        else if (h->model_id == 1) hipLaunchKernelGGL((k_cl_infeas<1>), dim3(B), dim3(64), 0, h->st, a, h->scp_upd, h->pinf);
        else hipLaunchKernelGGL((k_cl_infeas<2>), dim3(B), dim3(64), 0, h->st, a, h->scp_upd, h->pinf);
        if (h->log_steps > 0) {
            ClLogArgs la{h->cl_stepno, h->cl_done, h->B, d.N, d.nx, d.nu, h->log_steps, 0, h->Xn, h->Un, h->backoff_x, h->backoff_u, h->scp_success, h->scp_iters,
                         h->pinf, h->lg_x, h->lg_u, h->lg_bx, h->lg_bu, h->lg_state, h->lg_u0, h->lg_pinf, h->lg_succ, h->lg_it};
            hipLaunchKernelGGL(k_cl_log, dim3(1024), dim3(256), 0, h->st, la);
        }
        if (h->model_id == 0) hipLaunchKernelGGL((k_cl_shift_plant<0>), dim3(gb), dim3(64), 0, h->st, a, 0, 1, h->cl_done, h->cl_stepno, dW);
        else if (h->model_id == 1) hipLaunchKernelGGL((k_cl_shift_plant<1>), dim3(gb), dim3(64), 0, h->st, a, 0, 1, h->cl_done, h->cl_stepno, dW);
        else hipLaunchKernelGGL((k_cl_shift_plant<2>), dim3(gb), dim3(64), 0, h->st, a, 0, 1, h->cl_done, h->cl_stepno, dW);
        hipLaunchKernelGGL(k_cl_advance, dim3(gbi), dim3(256), 0, h->st, B, h->cl_done, h->cl_stepno, h->call_ids, h->cl_begin);
        HIPCHK(hipGetLastError());
    }
    tl_take(h);
    h->cl_steps = steps;
    h->call_id += steps;
    if (rounds_out) *rounds_out = rounds;
    return 0;
}

// ---- QP-level boundary (osqp_generated look-alike, batched) -------------------------------------------------
extern "C" int slsqp_qp_update_data_mat(slsqp_handle *h, const double *P_x, const double *A_x, int loc) {
    hipSetDevice(h->dev);
    const slsqp_dims &d = h->d;
    int n, m, nnzP, nnzA;
    slsqp_qp_nnz(&d, &n, &m, &nnzP, &nnzA);
    if (P_x) {   // diagonal of 2*blkdiag(Q,R,...,Qf): take instance 0 (weights are batch-constant)
        std::vector<double> p0(n);
        HIPCHK(hipMemcpy(p0.data(), P_x, sizeof(double) * n, loc == SLSQP_HOST ? hipMemcpyHostToHost : hipMemcpyDeviceToHost));
        std::vector<double> cur(2 * d.nx + d.nu);
        for (int i = 0; i < d.nx; i++) cur[i] = 0.5 * p0[i];
        for (int i = 0; i < d.nu; i++) cur[d.nx + i] = 0.5 * p0[d.nx + i];
        for (int i = 0; i < d.nx; i++) cur[d.nx + d.nu + i] = 0.5 * p0[(size_t)h->nz * d.N + i];
        HIPCHK(hipMemcpy(h->cst, cur.data(), sizeof(double) * cur.size(), hipMemcpyHostToDevice));
        h->have_costs = true;
    }
    if (A_x) {
        const double *src = A_x;
        if (loc == SLSQP_HOST) {
            double *tmp = stage_buf(h, sizeof(double) * (size_t)h->B * nnzA);
            if (!tmp) return -1;
            HIPCHK(hipMemcpyAsync(tmp, A_x, sizeof(double) * (size_t)h->B * nnzA, hipMemcpyHostToDevice, h->st));
            src = tmp;
        }
        hipLaunchKernelGGL(k_gather_csc, dim3(1024), dim3(256), 0, h->st, h->B, d.N * d.nx * d.nx, nnzA, h->mapA, src, h->A);
        hipLaunchKernelGGL(k_gather_csc, dim3(1024), dim3(256), 0, h->st, h->B, d.N * d.nx * d.nu, nnzA, h->mapB, src, h->Bm);
        HIPCHK(hipStreamSynchronize(h->st));
    }
    return 0;
}

extern "C" int slsqp_qp_update_data_vec(slsqp_handle *h, const double *q, const double *l, const double *u, int loc) {
    hipSetDevice(h->dev);
    const size_t B = h->B;
    if (q && put(h, h->q, q, sizeof(double) * B * h->n, loc)) return -1;
    if (l && u) {
        const double *dl = l, *du = u;
        if (loc == SLSQP_HOST) {
            double *tmp = stage_buf(h, sizeof(double) * 2 * B * h->m);
            if (!tmp) return -1;
            HIPCHK(hipMemcpyAsync(tmp, l, sizeof(double) * B * h->m, hipMemcpyHostToDevice, h->st));
            HIPCHK(hipMemcpyAsync(tmp + B * h->m, u, sizeof(double) * B * h->m, hipMemcpyHostToDevice, h->st));
            dl = tmp; du = tmp + B * h->m;
        }
        hipLaunchKernelGGL(k_split_lu, dim3(1024), dim3(256), 0, h->st, h->B, h->mb, h->d.nx, dl, du, h->lbg, h->ubg, h->x0val);
        HIPCHK(hipStreamSynchronize(h->st));
    }
    h->have_dyn = true;
    return 0;
}

extern "C" int slsqp_qp_solve(slsqp_handle *h, double *x, double *y, int *status, int *iters, int loc, const slsqp_opts *opts) {
    hipSetDevice(h->dev);
    if (h->general_G) return fail("general G: only the sweep-level boundary (slsqp_sweep) is available; the QP solver needs box constraints G = [I;-I]");
    if (!h->have_costs) return fail("costs not set");
    slsqp_opts o;
    if (opts) o = *opts; else slsqp_default_opts(&o);
    HIPCHK(hipEventRecord(h->ev[0], h->st));
    if (launch_qp(h, nullptr, &o, o.warm_start ? 1 : 0)) return -1;
    HIPCHK(hipEventRecord(h->ev[1], h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    harvest_kernel_events(h);
    h->t_total = h->t_qp = ev_ms(h->ev[0], h->ev[1]); h->t_sweep = 0;
    const hipMemcpyKind kd = loc == SLSQP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (x) HIPCHK(hipMemcpyAsync(x, h->primal, sizeof(double) * (size_t)h->B * h->n, kd, h->st));
    if (y) {
        double *tmp = stage_buf(h, sizeof(double) * (size_t)h->B * h->m);
        if (!tmp) return -1;
        hipLaunchKernelGGL(k_join_y, dim3(1024), dim3(256), 0, h->st, h->B, h->mb, h->d.nx, h->dual, h->pin_dual, tmp);
        HIPCHK(hipMemcpyAsync(y, tmp, sizeof(double) * (size_t)h->B * h->m, kd, h->st));
    }
    if (status) HIPCHK(hipMemcpyAsync(status, h->status, sizeof(int) * h->B, kd, h->st));
    if (iters) HIPCHK(hipMemcpyAsync(iters, h->iters, sizeof(int) * h->B, kd, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    return 0;
}

// ---- sweep-level boundary -----------------------------------------------------------------------------------
extern "C" int slsqp_sweep(slsqp_handle *h, const double *eta, const double *eta_f, double *K, double *beta, double *beta_f, double *backoff,
                           double *backoff_f, int loc) {
    hipSetDevice(h->dev);
    if (!h->have_costs) return fail("costs not set");
    const slsqp_dims &d = h->d;
    const size_t B = h->B;
    if (put(h, h->eta, eta, sizeof(double) * B * d.N * d.N * d.ni, loc)) return -1;
    if (put(h, h->eta_f, eta_f, sizeof(double) * B * (d.N + 1) * d.ni_f, loc)) return -1;
    hipLaunchKernelGGL(k_fill_int, dim3((h->B + 255) / 256), dim3(256), 0, h->st, h->stale, 8, h->B);   // eta given by the caller, K and beta written for every instance
    HIPCHK(hipEventRecord(h->ev[0], h->st));
    if (launch_sweep(h, nullptr, h->eta, h->eta_f, 1e-10)) return -1;
    HIPCHK(hipEventRecord(h->ev[1], h->st));
    TightenArgs ta{h->B, d.N, d.nx, d.nu, d.ni, d.ni_f, h->beta, h->beta_f, h->g, h->gf_raw, h->c, nullptr, h->backoff, h->backoff_f, h->backoff_x, h->backoff_u, h->ubg, 0, h->ct_part, h->cost_tube};
    hipLaunchKernelGGL(k_tighten, dim3(h->B), dim3(128), 0, h->st, ta);
    HIPCHK(hipEventRecord(h->ev[2], h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    h->t_sweep = ev_ms(h->ev[0], h->ev[1]); h->t_total = ev_ms(h->ev[0], h->ev[2]); h->t_qp = 0;
    const hipMemcpyKind kd = loc == SLSQP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (K) HIPCHK(hipMemcpyAsync(K, h->K, sizeof(double) * B * d.N * (d.N + 1) * d.nu * d.nx, kd, h->st));
    if (beta) HIPCHK(hipMemcpyAsync(beta, h->beta, sizeof(double) * B * d.N * d.N * d.ni, kd, h->st));
    if (beta_f) HIPCHK(hipMemcpyAsync(beta_f, h->beta_f, sizeof(double) * B * (d.N + 1) * d.ni_f, kd, h->st));
    if (backoff) HIPCHK(hipMemcpyAsync(backoff, h->backoff, sizeof(double) * B * d.N * d.ni, kd, h->st));
    if (backoff_f) HIPCHK(hipMemcpyAsync(backoff_f, h->backoff_f, sizeof(double) * B * d.ni_f, kd, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    return 0;
}


// ---- self-test of the wave-level building blocks (wave_la.hpp) ---------------------------------------------------
// One wave runs one primitive on operands the caller supplies; tests/test_gpu_parity.py compares with numpy.  `in` / `out` are packed
// row-major matrices in the order given per case below; device scratch is allocated per call (diagnostic entry point, not a hot path).
template <int NX, int NU>
__global__ __launch_bounds__(64) void k_selftest(int which, const double *in, double *out, int n_in, int n_out) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    constexpr int MM = NX * NX, NB = NX * NU;
    double *I0 = sm, *O0 = sm + n_in;
    for (int o = lane; o < n_in; o += 64) I0[o] = in[o];
    for (int o = lane; o < n_out; o += 64) O0[o] = 0.0;
    wla::wsync();
    if (which == 0) wla::gemm_mfma<NX, NX, NX, false, false>(I0, NX, I0 + MM, NX, O0, NX, lane);                                   // A B
    else if (which == 1) wla::gemm_mfma<NX, NX, NX, true, false>(I0, NX, I0 + MM, NX, O0, NX, lane);                                // A' B
    else if (which == 2) wla::gemm_mfma<NU, NX, NX, true, false>(I0, NU, I0 + NB, NX, O0, NX, lane);                                // Bm' S   (in: Bm NX x NU, S NX x NX)
    else if (which == 3) wla::gemm_mfma<NX, NX, NX, false, false, false, true>(I0, NX, I0 + MM, NX, O0, NX, lane, nullptr, 0, I0 + 2 * MM);   // A diag(s) B
    else if (which == 4) wla::gemm_mfma<NX, NX, NU, false, false, true>(I0, NU, I0 + NB, NX, O0, NX, lane, I0 + 2 * NB, NX);        // D + Bm K  (in: Bm, K NU x NX, D NX x NX)
    else if (which == 5) wla::gemm_mfma_pair<NU, NX, NX, NX>(I0, NX, I0 + NB, NX, I0 + NB + MM, NX, O0, NX, O0 + NB, NX, lane);     // K P | A P (in: K NU x NX, A, P)
    else if (which == 6) { const int f = wla::spd_inv_gj<NX>(I0, NX, O0, NX, (double *)nullptr, lane); if (lane == 0) O0[MM] = (double)f; }   // inverse of the SPD matrix whose lower triangle is given
    else if (which == 8) {   // the same inverse by the matrix-core sweep; its block-packed copy (4 x gj_blocks doubles) follows the flag in the output
        const int f = wla::spd_inv_gj_mfma<NX>(I0, NX, O0, NX, O0 + MM + 1, lane); if (lane == 0) O0[MM] = (double)f;
    }
    else if (which == 7) {   // lower(Y) = A diag(pix) A' + B diag(piu) B' - T (A diag(pix))' + diag(d) + delta    (in: A, B, T, pix, piu, d)
        const double *A = I0, *Bm = I0 + MM, *T = I0 + MM + NB, *pix = T + MM, *piu = pix + NX, *d = piu + NU;
        if constexpr (NX >= 5) wla::build_Y_mfma<NX, NU>(A, pix, Bm, piu, T, true, d, 1e-13, O0, lane);
    }
    wla::wsync();
    for (int o = lane; o < n_out; o += 64) out[o] = O0[o];
}

extern "C" int slsqp_selftest(int nx, int nu, int which, const double *in, int n_in, double *out, int n_out) {
    if (n_in <= 0 || n_out <= 0 || n_in + n_out > 4000) return fail("selftest: operand sizes");
    double *din = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc(&din, sizeof(double) * n_in));
    if (hipMalloc(&dout, sizeof(double) * n_out) != hipSuccess) { hipFree(din); return fail("selftest: hipMalloc"); }
    int rc = 0;
    if (hipMemcpy(din, in, sizeof(double) * n_in, hipMemcpyHostToDevice) != hipSuccess) rc = -1;
    const size_t lds = sizeof(double) * (size_t)(n_in + n_out);
    if (!rc) {
        if (nx == 17 && nu == 4) hipLaunchKernelGGL((k_selftest<17, 4>), dim3(1), dim3(64), lds, 0, which, din, dout, n_in, n_out);
        else if (nx == 13 && nu == 4) hipLaunchKernelGGL((k_selftest<13, 4>), dim3(1), dim3(64), lds, 0, which, din, dout, n_in, n_out);
        else rc = -2;
    }
    if (!rc && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, dout, sizeof(double) * n_out, hipMemcpyDeviceToHost) != hipSuccess)) rc = -1;
    hipFree(din); hipFree(dout);
    if (rc == -2) return fail("selftest: (nx, nu) must be (17, 4) or (13, 4)");
    return rc ? fail("selftest: HIP error") : 0;
}
