// slsqp_kernels.hpp -- device kernels of the batched fast-SLS QP path for gfx950 (MI355X).
//
// Mapping: ONE 64-lane wavefront per MPC instance (QP kernel) or per (instance, disturbance column)
// (sweep kernel); workgroups are a single wave, so thousands of independent horizon recursions are in
// flight and the scheduler interleaves them on every SIMD.  Stage blocks (A_k, B_k, Cholesky factors)
// are staged HBM/L2 -> LDS with coalesced loads; small dense algebra runs out of LDS (wave_la.hpp).
//
// Reference behaviour restated here (citations relative to antoineleeman/robust-nonlinear-mpc):
//   k_qp          the QP of QP.solve                 solver/qp_jit.py:362-402 (data layout :77-192)
//   k_eta         evaluate_dual_eta                  solver/fast_SLS_jit.py:475-487
//   k_sweep       _backward_solve_numba/_propagate/_backoff_from_phi (beta part)  :43-158
//   k_tighten     _backoff_from_phi (sums) + update_tightening tail             :160-188, :556-569
//   k_conv        check_convergence_socp             solver/fast_SLS_jit.py:581-600
//   k_set_bounds  QP.update_dynamics + offset_constraints  solver/qp_jit.py:268-273, 595-610
#pragma once
#include <hip/hip_runtime.h>

#include "wave_la.hpp"

namespace slsqp {

constexpr double BIGB = 1e19;   // |bound| above this is "infinite" (the reference maps +-inf to +-1e20, qp_jit.py:382)
constexpr double EPS_PIN = 1e-10;
constexpr int ST_INIT = -1;

struct Costs {  // batch-constant diagonal weights (device pointers)
    const double *Qd, *Rd, *Qfd;        // diag of Q, R, Qf (P = 2*blkdiag)
    const double *Qregd, *Rregd, *Qregfd;
};

// ------------------------------------------------------------------------------------------------
// row layout helpers (reference: qp_jit.py:101-123)
// ------------------------------------------------------------------------------------------------
template <int NX, int NU>
struct Lay {
    static constexpr int NZ = NX + NU, NI = 2 * NZ, NIF = 2 * NX, SR = NX + NI;  // SR = rows per stage
    __host__ __device__ static int n(int N) { return NZ * N + NX; }
    __host__ __device__ static int mb(int N) { return N * SR + NIF; }
};

// ------------------------------------------------------------------------------------------------
// QP kernel: primal-dual interior point on the block-tridiagonal normal equations + active-set polish
// ------------------------------------------------------------------------------------------------
struct QpArgs {
    int B, N;
    const double *A, *Bm;     // (B,N,NX,NX) (B,N,NX,NU)
    const double *q;          // (B,n)
    const double *ubg, *lbg;  // (B,mb) reference row layout
    const double *x0val;      // (B,NX) value x_0 is pinned to
    const int *run;           // (B) 1 = solve this instance (NULL = all)
    Costs cst;
    double *Linv;             // scratch (B,N,NX,NX)
    double *primal;           // (B,n)
    double *dual;             // (B,mb)
    double *cost;             // (B)
    double *pin_dual;         // (B,NX) multipliers of the x0-pin rows (may be NULL)
    double *kkt;              // (B,8) [0..3] accepted solution: stationarity, box violation, multiplier-sign violation, mu;
                              //       [4..7] last polish attempt: stationarity, box, sign, factorisation-failed flag
    int *status, *iters;      // (B)
    int max_iter;
    double eps;
};

template <int NX, int NU>
struct NeCtx {
    static constexpr int NZ = NX + NU;
    double *sA, *sB, *sBs, *sM1, *sY, *sLa, *sLb, *sL1, *sCol;
    double *sPi, *sV, *sG, *sW, *sT1, *sT2, *sT3;
    const double *gA, *gB;        // this instance's A (N,NX,NX), B (N,NX,NU)
    const double *gUb, *gLb;      // this instance's ubg / lbg rows (eq rhs = centre of the dynamics rows' box)
    double *gLinv;                // scratch: Linv_k (N,NX,NX)
    int N, lane;
};

// Forward sweep over the horizon of the block-tridiagonal normal equations  Y nu = b,  Y = E Pi E':
//   optional (re)factorisation  L_kk L_kk' = Y_kk - L_{k,k-1} L_{k,k-1}'  (explicit inverse Linv_k kept),
//   rhs b_k = E_k v - eflag * e_k, block forward substitution w_k = Linv_k (b_k - L_{k,k-1} w_{k-1}).
// One call site only (the caller is a phase machine): keeps code size and register pressure down.
template <int NX, int NU>
__device__ __forceinline__ int ne_forward(const NeCtx<NX, NU> c, bool factor, double eflag, double delta) {
    constexpr int NZ = NX + NU, SR = NX + 2 * NZ;
    const int lane = c.lane;
    int fail = 0;
    double *Lcur = c.sLa, *Lprev = c.sLb;
    for (int k = 0; k < c.N; k++) {
        const double *Ak = c.gA + (size_t)k * NX * NX, *Bk = c.gB + (size_t)k * NX * NU;
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) c.sA[o] = Ak[o];
#pragma unroll
        for (int o = lane; o < NX * NU; o += 64) c.sB[o] = Bk[o];
        if (!factor) {
            const double *Lg = c.gLinv + (size_t)k * NX * NX;
#pragma unroll
            for (int o = lane; o < NX * NX; o += 64) Lcur[o] = Lg[o];
        }
        double ek = 0.0;
        if (lane < NX && eflag != 0.0) ek = 0.5 * (c.gUb[k * SR + lane] + c.gLb[k * SR + lane]);
        wla::wsync();
        if (factor) {
            const double *pix = c.sPi + k * NZ, *piu = pix + NX, *pixn = c.sPi + (k + 1) * NZ;
#pragma unroll
            for (int o = lane; o < NX * NX; o += 64) c.sM1[o] = (k > 0) ? c.sA[o] * pix[o % NX] : 0.0;
#pragma unroll
            for (int o = lane; o < NX * NU; o += 64) c.sBs[o] = c.sB[o] * piu[o % NU];
            wla::wsync();
            // Y = M1 A' + Bs B' + diag(pi_x,k+1) + delta
#pragma unroll
            for (int o0 = 0; o0 < NX * NX; o0 += 64) {
                const int o = o0 + lane;
                if (o < NX * NX) {
                    const int i = o / NX, j = o % NX;
                    double s = (i == j) ? pixn[i] + delta : 0.0;
#pragma unroll
                    for (int m = 0; m < NX; m++) s = fma(c.sM1[i * NX + m], c.sA[j * NX + m], s);
#pragma unroll
                    for (int m = 0; m < NU; m++) s = fma(c.sBs[i * NU + m], c.sB[j * NU + m], s);
                    c.sY[o] = s;
                }
            }
            if (k > 0) wla::gemm<NX, NX, NX, false, true>(c.sM1, NX, Lprev, NX, c.sL1, NX, -1.0, 0.0, lane);  // L_{k,k-1} = -M1 Linv_{k-1}'
            wla::wsync();
            if (k > 0) {
                wla::gemm<NX, NX, NX, false, true>(c.sL1, NX, c.sL1, NX, c.sY, NX, -1.0, 1.0, lane);
                wla::wsync();
            }
            fail |= wla::chol_inv<NX>(c.sY, NX, Lcur, NX, c.sCol, lane);
            double *Lg = c.gLinv + (size_t)k * NX * NX;
#pragma unroll
            for (int o = lane; o < NX * NX; o += 64) Lg[o] = Lcur[o];
        }
        // rhs
        const double *vx = c.sV + k * NZ, *vu = vx + NX, *vxn = c.sV + (k + 1) * NZ;
        double b = 0.0;
        if (lane < NX) {
            b = -vxn[lane] - eflag * ek;
#pragma unroll
            for (int m = 0; m < NX; m++) b = fma(c.sA[lane * NX + m], vx[m], b);
#pragma unroll
            for (int m = 0; m < NU; m++) b = fma(c.sB[lane * NU + m], vu[m], b);
        }
        if (k > 0) {
            // -L_{k,k-1} w_{k-1} = A_k (pi_x,k .* (Linv_{k-1}' w_{k-1}))
            const double t = wla::matvec_row<NX, NX, true>(Lprev, NX, c.sW + (k - 1) * NX, lane);
            if (lane < NX) c.sT1[lane] = t * c.sPi[k * NZ + lane];
            wla::wsync();
            if (lane < NX) {
#pragma unroll
                for (int m = 0; m < NX; m++) b = fma(c.sA[lane * NX + m], c.sT1[m], b);
            }
        }
        if (lane < NX) c.sT2[lane] = b;
        wla::wsync();
        const double w = wla::matvec_row<NX, NX, false>(Lcur, NX, c.sT2, lane);
        if (lane < NX) c.sW[k * NX + lane] = w;
        wla::wsync();
        double *t = Lcur; Lcur = Lprev; Lprev = t;
    }
    return fail;
}

// Backward sweep: nu_k = Linv_k' (w_k - L_{k+1,k}' nu_{k+1}) (overwrites w in sW) and G = E' nu (n-vector in sG).
template <int NX, int NU>
__device__ __forceinline__ void ne_backward(const NeCtx<NX, NU> c) {
    constexpr int NZ = NX + NU;
    const int lane = c.lane;
    if (lane < NX) c.sT3[lane] = 0.0;  // A_{k+1}' nu_{k+1}
    wla::wsync();
    for (int k = c.N - 1; k >= 0; k--) {
        const double *Ak = c.gA + (size_t)k * NX * NX, *Bk = c.gB + (size_t)k * NX * NU, *Lg = c.gLinv + (size_t)k * NX * NX;
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) { c.sA[o] = Ak[o]; c.sLa[o] = Lg[o]; }
#pragma unroll
        for (int o = lane; o < NX * NU; o += 64) c.sB[o] = Bk[o];
        if (lane < NX) c.sT1[lane] = c.sPi[(k + 1) * NZ + lane] * c.sT3[lane];
        wla::wsync();
        const double tmp = wla::matvec_row<NX, NX, false>(c.sLa, NX, c.sT1, lane);
        if (lane < NX) c.sT2[lane] = c.sW[k * NX + lane] + tmp;
        wla::wsync();
        const double nu = wla::matvec_row<NX, NX, true>(c.sLa, NX, c.sT2, lane);
        if (lane < NX) {
            c.sW[k * NX + lane] = nu;
            c.sG[(k + 1) * NZ + lane] = c.sT3[lane] - nu;
        }
        wla::wsync();
        const double ga = wla::matvec_row<NX, NX, true>(c.sA, NX, c.sW + k * NX, lane);
        const double gb = wla::matvec_row<NU, NX, true>(c.sB, NU, c.sW + k * NX, lane);
        if (lane < NX) c.sT3[lane] = ga;
        if (lane < NU) c.sG[k * NZ + NX + lane] = gb;
        wla::wsync();
    }
    if (lane < NX) c.sG[lane] = c.sT3[lane];
    wla::wsync();
}

template <int NX, int NU>
__host__ __device__ constexpr int qp_lds_doubles(int N) {
    // sA sM1 sY sLa sLb sL1 (6 NX^2) + sB sBs (2 NX NU) + sCol,sT1..3 (4 NX) + sPi sV sG (3 n) + sW sNu sNuP (3 N NX)
    return 6 * NX * NX + 2 * NX * NU + 4 * NX + 3 * ((NX + NU) * N + NX) + 3 * N * NX + 8;
}

// per-element constants of the stage-ordered primal vector, re-read from L2-resident inputs where needed
struct Elem { double hi, lo, q, pd; bool fu, fl, fr; };
template <int NX, int NU>
__device__ __forceinline__ Elem elem_of(int e, int n, int N, const double *ub, const double *qg, const Costs &cst) {
    constexpr int NZ = NX + NU, SR = NX + 2 * NZ;
    Elem r;
    r.hi = 1e20; r.lo = -1e20; r.q = 0.0; r.pd = 1.0; r.fu = r.fl = r.fr = false;
    if (e < n) {
        const int k = e / NZ, i = e % NZ;
        r.q = qg[e];
        if (k < N) {
            r.pd = 2.0 * (i < NX ? cst.Qd[i] : cst.Rd[i - NX]);
            r.hi = ub[k * SR + NX + i];
            r.lo = -ub[k * SR + NX + NZ + i];
        } else {
            r.pd = 2.0 * cst.Qfd[i];
            r.hi = ub[N * SR + i];
            r.lo = -ub[N * SR + NX + i];
        }
        r.fr = e >= NX;
        r.fu = r.fr && r.hi < BIGB;
        r.fl = r.fr && r.lo > -BIGB;
    }
    return r;
}

// QP kernel.  Algorithm (per instance, one wave):
//   1. equality-constrained optimum (bounds ignored) as starting point,
//   2. Mehrotra predictor-corrector interior point; every Newton system is reduced to the block-tridiagonal
//      normal equations  (E Pi E') dnu = rhs,  Pi = (P + Sigma)^-1 diagonal, factorised stage by stage,
//   3. active-set polish (the OSQP-polish idea, qp_jit.py:546 `polishing=True`): fix the variables the interior
//      point identifies as active, re-solve the KKT system exactly with 2 refinement steps, accept only if the
//      KKT certificate (stationarity, box feasibility, multiplier signs) holds to 1e-9.
template <int NX, int NU, int TV>
__global__ __launch_bounds__(64) void k_qp(QpArgs a) {
    using L = Lay<NX, NU>;
    constexpr int NZ = L::NZ, SR = L::SR;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= a.B) return;
    if (a.run && !a.run[b]) return;
    const int N = a.N, n = L::n(N), mb = L::mb(N);
    extern __shared__ double sm[];
    NeCtx<NX, NU> c;
    double *p = sm;
    c.sA = p; p += NX * NX; c.sM1 = p; p += NX * NX; c.sY = p; p += NX * NX; c.sLa = p; p += NX * NX;
    c.sLb = p; p += NX * NX; c.sL1 = p; p += NX * NX; c.sB = p; p += NX * NU; c.sBs = p; p += NX * NU;
    c.sCol = p; p += NX; c.sT1 = p; p += NX; c.sT2 = p; p += NX; c.sT3 = p; p += NX;
    c.sPi = p; p += n; c.sV = p; p += n; c.sG = p; p += n;
    c.sW = p; p += N * NX; double *sNu = p; p += N * NX; double *sNuP = p; p += N * NX;
    c.gA = a.A + (size_t)b * N * NX * NX; c.gB = a.Bm + (size_t)b * N * NX * NU; c.gLinv = a.Linv + (size_t)b * N * NX * NX;
    const double *ub = a.ubg + (size_t)b * mb, *lb = a.lbg + (size_t)b * mb, *qg = a.q + (size_t)b * n;
    c.gUb = ub; c.gLb = lb; c.N = N; c.lane = lane;
    const Costs cst = a.cst;

    int status = ST_INIT;
    {   // x0 pin vs its own box (the reference applies both the pin rows and the stage-0 inequality rows)
        double viol = 0.0;
        if (lane < NX) {
            const double xv = a.x0val[(size_t)b * NX + lane];
            viol = fmax(xv - ub[NX + lane], -ub[NX + NZ + lane] - xv);
        }
        if (wla::wave_max(viol) > 1e-9) status = 2;
    }
    double z[TV], su[TV], sl[TV], lu[TV], ll[TV], gc[TV], cu[TV], cl[TV];
    double qscale = 0.0, mtot = 0.0;
#pragma unroll
    for (int t = 0; t < TV; t++) {
        const int e = t * 64 + lane;
        const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
        qscale = fmax(qscale, fabs(el.q));
        mtot += (el.fu ? 1.0 : 0.0) + (el.fl ? 1.0 : 0.0);
        // phase INIT rhs: v = z0 - Pi (P z0 + q) with z0 = [x0; 0]
        const double z0 = (e < NX) ? a.x0val[(size_t)b * NX + e] : 0.0;
        const double pi = el.fr ? 1.0 / el.pd : 0.0;
        if (e < n) { c.sPi[e] = pi; c.sV[e] = z0 - pi * (el.pd * z0 + el.q); }
        z[t] = z0; su[t] = sl[t] = 1.0; lu[t] = ll[t] = 0.0; gc[t] = 0.0; cu[t] = cl[t] = 0.0;
    }
    qscale = fmax(1.0, wla::wave_max(qscale));
    mtot = fmax(1.0, wla::wave_sum(mtot));
    const double tol = a.eps * qscale;
    for (int o = lane; o < N * NX; o += 64) { sNu[o] = 0.0; sNuP[o] = 0.0; }
    wla::wsync();

    enum { P_INIT = 0, P_PRED = 1, P_CORR = 2, P_POL0 = 3, P_POL1 = 4, P_POL2 = 5, P_DONE = 6 };
    int phase = (status == ST_INIT) ? P_INIT : P_DONE;
    int it = 0, chol_fail = 0, pol_fail = 0, pol_round = 0;
    double mu = 0.0, res = 0.0, smu = 0.0, alpha = 0.0;
    double kst = 0.0, kbox = 0.0, ksign = 0.0, pst = -1.0, pbox = -1.0, psign = -1.0;
    bool polished = false;
    unsigned actU = 0u, actL = 0u;   // bit t: element t*64+lane is held at its upper / lower bound by the polish

    while (phase != P_DONE) {
        const bool factor = (phase == P_INIT || phase == P_PRED || phase == P_POL0);
        const double eflag = (phase == P_INIT || phase >= P_POL0) ? 1.0 : 0.0;
        const int f = ne_forward<NX, NU>(c, factor, eflag, phase == P_POL0 ? 1e-10 : 0.0);
        ne_backward<NX, NU>(c);
        if (phase == P_POL0) pol_fail = f; else chol_fail |= f;

        bool start_iter = false;
        if (phase == P_INIT) {
            for (int o = lane; o < N * NX; o += 64) sNu[o] = c.sW[o];
#pragma unroll
            for (int t = 0; t < TV; t++) {
                const int e = t * 64 + lane;
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                if (e < n) {
                    gc[t] = c.sG[e];
                    if (el.fr) z[t] = -(el.q + gc[t]) / el.pd;
                    if (el.fu) { su[t] = fmax(el.hi - z[t], 1.0); lu[t] = 1.0; }
                    if (el.fl) { sl[t] = fmax(z[t] - el.lo, 1.0); ll[t] = 1.0; }
                }
            }
            start_iter = true;
        } else if (phase == P_PRED) {
            // affine step -> centring parameter -> corrector rhs
            double amin = 1.0;
#pragma unroll
            for (int t = 0; t < TV; t++) {
                const int e = t * 64 + lane;
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double dz = (e < n) ? c.sV[e] - c.sPi[e] * c.sG[e] : 0.0;
                const double ru = el.fu ? z[t] + su[t] - el.hi : 0.0, rl = el.fl ? el.lo - z[t] + sl[t] : 0.0;
                const double Wu = el.fu ? lu[t] / su[t] : 0.0, Wl = el.fl ? ll[t] / sl[t] : 0.0;
                const double tu = el.fu ? (-su[t] * lu[t] + lu[t] * ru) / su[t] : 0.0;
                const double tl = el.fl ? (-sl[t] * ll[t] + ll[t] * rl) / sl[t] : 0.0;
                const double dsu = el.fu ? -ru - dz : 0.0, dsl = el.fl ? -rl + dz : 0.0;
                const double dlu = tu + Wu * dz, dll = tl - Wl * dz;
                if (el.fu) { if (dsu < 0) amin = fmin(amin, -su[t] / dsu); if (dlu < 0) amin = fmin(amin, -lu[t] / dlu); }
                if (el.fl) { if (dsl < 0) amin = fmin(amin, -sl[t] / dsl); if (dll < 0) amin = fmin(amin, -ll[t] / dll); }
                cu[t] = dsu; cl[t] = dsl;       // hold the affine slack steps; multiplied by dlambda below
                gc[t] += 0.0;
                // stash dlu/dll in sG/sV? -> recompute below from dz (cheap)
            }
            const double aaff = wla::wave_min(amin);
            double s1 = 0.0;
#pragma unroll
            for (int t = 0; t < TV; t++) {
                const int e = t * 64 + lane;
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double dz = (e < n) ? c.sV[e] - c.sPi[e] * c.sG[e] : 0.0;
                const double ru = el.fu ? z[t] + su[t] - el.hi : 0.0, rl = el.fl ? el.lo - z[t] + sl[t] : 0.0;
                const double Wu = el.fu ? lu[t] / su[t] : 0.0, Wl = el.fl ? ll[t] / sl[t] : 0.0;
                const double tu = el.fu ? (-su[t] * lu[t] + lu[t] * ru) / su[t] : 0.0;
                const double tl = el.fl ? (-sl[t] * ll[t] + ll[t] * rl) / sl[t] : 0.0;
                const double dlu = tu + Wu * dz, dll = tl - Wl * dz;
                s1 += (el.fu ? (su[t] + aaff * cu[t]) * (lu[t] + aaff * dlu) : 0.0) + (el.fl ? (sl[t] + aaff * cl[t]) * (ll[t] + aaff * dll) : 0.0);
                cu[t] *= dlu; cl[t] *= dll;     // second-order terms ds*dlambda
            }
            const double muaff = wla::wave_sum(s1) / mtot;
            double sig = muaff / mu; sig = sig * sig * sig;
            smu = sig * mu;
            wla::wsync();
#pragma unroll
            for (int t = 0; t < TV; t++) {
                const int e = t * 64 + lane;
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double ru = el.fu ? z[t] + su[t] - el.hi : 0.0, rl = el.fl ? el.lo - z[t] + sl[t] : 0.0;
                const double rd = el.fr ? el.pd * z[t] + el.q + gc[t] + lu[t] - ll[t] : 0.0;
                const double tu = el.fu ? (-(su[t] * lu[t] + cu[t] - smu) + lu[t] * ru) / su[t] : 0.0;
                const double tl = el.fl ? (-(sl[t] * ll[t] + cl[t] - smu) + ll[t] * rl) / sl[t] : 0.0;
                if (e < n) c.sV[e] = -c.sPi[e] * (rd + tu - tl);
            }
            phase = P_CORR;
        } else if (phase == P_CORR) {
            double amin = 1e300;
#pragma unroll
            for (int t = 0; t < TV; t++) {
                const int e = t * 64 + lane;
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double dz = (e < n) ? c.sV[e] - c.sPi[e] * c.sG[e] : 0.0;
                const double ru = el.fu ? z[t] + su[t] - el.hi : 0.0, rl = el.fl ? el.lo - z[t] + sl[t] : 0.0;
                const double Wu = el.fu ? lu[t] / su[t] : 0.0, Wl = el.fl ? ll[t] / sl[t] : 0.0;
                const double tu = el.fu ? (-(su[t] * lu[t] + cu[t] - smu) + lu[t] * ru) / su[t] : 0.0;
                const double tl = el.fl ? (-(sl[t] * ll[t] + cl[t] - smu) + ll[t] * rl) / sl[t] : 0.0;
                const double dsu = el.fu ? -ru - dz : 0.0, dsl = el.fl ? -rl + dz : 0.0;
                const double dlu = tu + Wu * dz, dll = tl - Wl * dz;
                if (el.fu) { if (dsu < 0) amin = fmin(amin, -su[t] / dsu); if (dlu < 0) amin = fmin(amin, -lu[t] / dlu); }
                if (el.fl) { if (dsl < 0) amin = fmin(amin, -sl[t] / dsl); if (dll < 0) amin = fmin(amin, -ll[t] / dll); }
            }
            alpha = fmin(1.0, 0.99 * wla::wave_min(amin));
#pragma unroll
            for (int t = 0; t < TV; t++) {
                const int e = t * 64 + lane;
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double g = (e < n) ? c.sG[e] : 0.0;
                const double dz = (e < n) ? c.sV[e] - c.sPi[e] * g : 0.0;
                const double ru = el.fu ? z[t] + su[t] - el.hi : 0.0, rl = el.fl ? el.lo - z[t] + sl[t] : 0.0;
                const double Wu = el.fu ? lu[t] / su[t] : 0.0, Wl = el.fl ? ll[t] / sl[t] : 0.0;
                const double tu = el.fu ? (-(su[t] * lu[t] + cu[t] - smu) + lu[t] * ru) / su[t] : 0.0;
                const double tl = el.fl ? (-(sl[t] * ll[t] + cl[t] - smu) + ll[t] * rl) / sl[t] : 0.0;
                const double dsu = el.fu ? -ru - dz : 0.0, dsl = el.fl ? -rl + dz : 0.0;
                const double dlu = tu + Wu * dz, dll = tl - Wl * dz;
                z[t] += alpha * dz; gc[t] += alpha * g;
                if (el.fu) { su[t] += alpha * dsu; lu[t] += alpha * dlu; }
                if (el.fl) { sl[t] += alpha * dsl; ll[t] += alpha * dll; }
            }
            for (int o = lane; o < N * NX; o += 64) sNu[o] += alpha * c.sW[o];
            it++;
            start_iter = true;
        } else {
            // polish phases: zn = v - Pi g  (v was z0 - Pi r or zn - Pi r1), accumulate g and nu
#pragma unroll
            for (int t = 0; t < TV; t++) {
                const int e = t * 64 + lane;
                const double g = (e < n) ? c.sG[e] : 0.0;
                cu[t] = (e < n) ? c.sV[e] - c.sPi[e] * g : 0.0;   // cu := zn
                cl[t] = (phase == P_POL0 ? 0.0 : cl[t]) + g;        // cl := accumulated E' nu
            }
            for (int o = lane; o < N * NX; o += 64) sNuP[o] = (phase == P_POL0 ? 0.0 : sNuP[o]) + c.sW[o];
            wla::wsync();
            if (phase != P_POL2) {
#pragma unroll
                for (int t = 0; t < TV; t++) {
                    const int e = t * 64 + lane;
                    const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                    if (e < n) {
                        const double pi = c.sPi[e];
                        const double r1 = (pi != 0.0) ? el.pd * cu[t] + el.q + cl[t] : 0.0;
                        c.sV[e] = cu[t] - pi * r1;
                    }
                }
                phase = phase + 1;
            } else {
                double vst = 0.0, vbox = 0.0, vsign = 0.0;
#pragma unroll
                for (int t = 0; t < TV; t++) {
                    const int e = t * 64 + lane;
                    const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                    const double gr = el.pd * cu[t] + el.q + cl[t];
                    const bool aU = (actU >> t) & 1u, aL = (actL >> t) & 1u;
                    if (el.fr && !aU && !aL) vst = fmax(vst, fabs(gr));
                    if (el.fu) vbox = fmax(vbox, cu[t] - el.hi);
                    if (el.fl) vbox = fmax(vbox, el.lo - cu[t]);
                    if (aU) vsign = fmax(vsign, gr);    // lambda_u = -gr must be >= 0
                    if (aL) vsign = fmax(vsign, -gr);   // lambda_l = +gr must be >= 0
                }
                vst = wla::wave_max(vst); vbox = wla::wave_max(vbox); vsign = wla::wave_max(vsign);
                const double ptol = 1e-9 * qscale;
                pst = vst; pbox = vbox; psign = vsign;
                if (!pol_fail && vst < ptol && vbox < ptol && vsign < ptol) {
                    polished = true; status = 0; kst = vst; kbox = vbox; ksign = vsign;
                    phase = P_DONE;
                } else if (!pol_fail && vst < ptol && pol_round < 6) {
                    // primal-dual active-set correction: release constraints whose multiplier has the wrong sign,
                    // add violated bounds, factorise again
                    pol_round++;
                    unsigned nU = 0u, nL = 0u;
#pragma unroll
                    for (int t = 0; t < TV; t++) {
                        const int e = t * 64 + lane;
                        const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                        const double gr = el.pd * cu[t] + el.q + cl[t];
                        bool aU = (actU >> t) & 1u, aL = (actL >> t) & 1u;
                        if (aU && gr > ptol) aU = false;
                        if (aL && -gr > ptol) aL = false;
                        if (!aU && !aL) {
                            if (el.fu && cu[t] > el.hi + ptol) aU = true;
                            else if (el.fl && cu[t] < el.lo - ptol) aL = true;
                        }
                        if (aU) nU |= (1u << t);
                        if (aL) nL |= (1u << t);
                        const double z0 = aU ? el.hi : (aL ? el.lo : cu[t]);
                        const double pi = (el.fr && !aU && !aL) ? 1.0 / el.pd : 0.0;
                        if (e < n) { c.sPi[e] = pi; c.sV[e] = z0 - pi * (el.pd * z0 + el.q); }
                    }
                    actU = nU; actL = nL;
                    phase = P_POL0;
                } else phase = P_DONE;
            }
        }

        if (start_iter) {
            // residuals, complementarity, termination test; then either predictor rhs or polish rhs
            double rmax = 0.0, musum = 0.0;
#pragma unroll
            for (int t = 0; t < TV; t++) {
                const int e = t * 64 + lane;
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double rd = el.fr ? el.pd * z[t] + el.q + gc[t] + lu[t] - ll[t] : 0.0;
                const double ru = el.fu ? z[t] + su[t] - el.hi : 0.0, rl = el.fl ? el.lo - z[t] + sl[t] : 0.0;
                rmax = fmax(rmax, fmax(fabs(rd), fmax(fabs(ru), fabs(rl))));
                musum += (el.fu ? su[t] * lu[t] : 0.0) + (el.fl ? sl[t] * ll[t] : 0.0);
            }
            res = wla::wave_max(rmax);
            mu = wla::wave_sum(musum) / mtot;
            kst = res;
            wla::wsync();
            if (!(res == res) || !(mu == mu) || res > 1e30) { status = 3; phase = P_DONE; }
            else if (res < tol && mu < tol) {
                status = 4;
                // polish rhs: active set, z0 with active entries on their bounds, Pi = 0 there
                actU = 0u; actL = 0u;
#pragma unroll
                for (int t = 0; t < TV; t++) {
                    const int e = t * 64 + lane;
                    const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                    const double lam = lu[t] - ll[t];
                    const bool aU = el.fu && (lam > el.hi - z[t]);
                    const bool aL = el.fl && !aU && (-lam > z[t] - el.lo);
                    if (aU) actU |= (1u << t);
                    if (aL) actL |= (1u << t);
                    const double z0 = aU ? el.hi : (aL ? el.lo : z[t]);
                    const double pi = (el.fr && !aU && !aL) ? 1.0 / el.pd : 0.0;
                    if (e < n) { c.sPi[e] = pi; c.sV[e] = z0 - pi * (el.pd * z0 + el.q); }
                }
                phase = P_POL0;
            } else if (it >= a.max_iter) { status = 1; phase = P_DONE; }
            else {
#pragma unroll
                for (int t = 0; t < TV; t++) {
                    const int e = t * 64 + lane;
                    const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                    const double rd = el.fr ? el.pd * z[t] + el.q + gc[t] + lu[t] - ll[t] : 0.0;
                    const double ru = el.fu ? z[t] + su[t] - el.hi : 0.0, rl = el.fl ? el.lo - z[t] + sl[t] : 0.0;
                    const double Wu = el.fu ? lu[t] / su[t] : 0.0, Wl = el.fl ? ll[t] / sl[t] : 0.0;
                    const double pi = el.fr ? 1.0 / (el.pd + Wu + Wl) : 0.0;
                    const double tu = el.fu ? (-su[t] * lu[t] + lu[t] * ru) / su[t] : 0.0;
                    const double tl = el.fl ? (-sl[t] * ll[t] + ll[t] * rl) / sl[t] : 0.0;
                    if (e < n) { c.sPi[e] = pi; c.sV[e] = -pi * (rd + tu - tl); }
                }
                phase = P_PRED;
            }
        }
        wla::wsync();
    }

    // ---- write-out (reference layouts: primal qp_jit.py:489-490, duals :493-501) ----
    double *pr = a.primal + (size_t)b * n, *du = a.dual + (size_t)b * mb;
    double csum = 0.0;
    const bool ok = (status == 0 || status == 4);   // on failure the previous primal/dual stay (fast_SLS_jit.py:461-464)
#pragma unroll
    for (int t = 0; t < TV; t++) {
        const int e = t * 64 + lane;
        if (e < n && ok) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            double zv = z[t], yu = el.fu ? lu[t] : 0.0, yl = el.fl ? ll[t] : 0.0, gv = gc[t];
            if (polished) {
                zv = cu[t]; gv = cl[t];
                const double gr = el.pd * zv + el.q + gv;
                yu = ((actU >> t) & 1u) ? fmax(-gr, 0.0) : 0.0;
                yl = ((actL >> t) & 1u) ? fmax(gr, 0.0) : 0.0;
            }
            pr[e] = zv;
            csum += 0.5 * el.pd * zv * zv + el.q * zv;
            const int k = e / NZ, i = e % NZ;
            if (k < N) { du[k * SR + NX + i] = yu; du[k * SR + NX + NZ + i] = yl; }
            else { du[N * SR + i] = yu; du[N * SR + NX + i] = yl; }
            if (e < NX && a.pin_dual) a.pin_dual[(size_t)b * NX + e] = -(el.pd * zv + el.q + gv);
        }
    }
    if (ok) { const double *nus = polished ? sNuP : sNu; for (int o = lane; o < N * NX; o += 64) du[(o / NX) * SR + (o % NX)] = nus[o]; }
    csum = wla::wave_sum(csum);
    (void)chol_fail;  // clamped pivots are tolerated when the certificate holds
    if (lane == 0) {
        if (ok) a.cost[b] = csum;
        a.status[b] = status;
        a.iters[b] = it;
        double *kk = a.kkt + (size_t)b * 8;
        kk[0] = kst; kk[1] = kbox; kk[2] = ksign; kk[3] = mu; kk[4] = pst; kk[5] = pbox; kk[6] = psign; kk[7] = (double)pol_fail;
    }
}

// ------------------------------------------------------------------------------------------------
// bounds of the un-tightened QP: QP.update_dynamics (qp_jit.py:268-273) + offset_constraints (:595-610)
// ------------------------------------------------------------------------------------------------
struct BoundsArgs { int B, N, NX, NI, NIF; const double *g, *gN, *c; double *ubg, *lbg; double eps; };
__global__ void k_set_bounds(BoundsArgs a) {
    const int SR = a.NX + a.NI, mb = a.N * SR + a.NIF;
    const size_t tot = (size_t)a.B * mb;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = idx / mb, r = idx % mb;
        double u, l;
        if (r < a.N * SR) {
            const int k = r / SR, i = r % SR;
            if (i < a.NX) { const double cv = a.c[((size_t)b * a.N + k) * a.NX + i]; u = -cv + a.eps; l = -cv - a.eps; }
            else { u = a.g[((size_t)b * a.N + k) * a.NI + (i - a.NX)] + a.eps; l = -1e20; }
        } else { u = a.gN[(size_t)b * a.NIF + (r - a.N * SR)] + a.eps; l = -1e20; }
        a.ubg[idx] = u; a.lbg[idx] = l;
    }
}

// ------------------------------------------------------------------------------------------------
// evaluate_dual_eta (fast_SLS_jit.py:475-487)
// ------------------------------------------------------------------------------------------------
struct EtaArgs { int B, N, NX, NI, NIF; const double *dual, *beta, *beta_f; const int *run; double *eta, *eta_f; double eps; };
__global__ void k_eta(EtaArgs a) {
    const int b = blockIdx.x;
    if (a.run && !a.run[b]) return;
    const int SR = a.NX + a.NI, mb = a.N * SR + a.NIF;
    const double *du = a.dual + (size_t)b * mb;
    const double *be = a.beta + (size_t)b * a.N * a.N * a.NI;
    double *et = a.eta + (size_t)b * a.N * a.N * a.NI;
    for (int o = threadIdx.x; o < a.N * a.N * a.NI; o += blockDim.x) {
        const int i = o % a.NI, j = (o / a.NI) % a.N, k = o / (a.NI * a.N);
        if (j <= k) et[o] = du[k * SR + a.NX + i] / (2.0 * sqrt(fmax(be[o], a.eps)));
    }
    const double *bf = a.beta_f + (size_t)b * (a.N + 1) * a.NIF;
    double *ef = a.eta_f + (size_t)b * (a.N + 1) * a.NIF;
    for (int o = threadIdx.x; o < (a.N + 1) * a.NIF; o += blockDim.x) {
        const int i = o % a.NIF;
        ef[o] = du[a.N * SR + i] / (2.0 * sqrt(fmax(bf[o], a.eps)));
    }
}

// ------------------------------------------------------------------------------------------------
// check_convergence_socp (fast_SLS_jit.py:581-600): state persists across calls
// ------------------------------------------------------------------------------------------------
struct ConvArgs { int B, n; const double *primal; double *prev; int *has_prev; const int *run; int *conv; double tol; };
__global__ void k_conv(ConvArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (a.run && !a.run[b]) { if (lane == 0) a.conv[b] = 0; return; }
    const double *p = a.primal + (size_t)b * a.n;
    double *q = a.prev + (size_t)b * a.n;
    double d = 0.0;
    for (int o = lane; o < a.n; o += 64) { d = fmax(d, fabs(p[o] - q[o])); q[o] = p[o]; }
    d = wla::wave_max(d);
    if (lane == 0) { a.conv[b] = (a.has_prev[b] && d <= a.tol) ? 1 : 0; a.has_prev[b] = 1; }
}

// ------------------------------------------------------------------------------------------------
// SLS sweep: one wave per (instance, disturbance column j)
// ------------------------------------------------------------------------------------------------
struct SweepArgs {
    int B, N, NW;
    const double *A, *Bm, *E;  // E (N+1,NX,NW) shared or (B,N+1,NX,NW)
    int E_per_instance;
    const double *eta, *eta_f;
    const int *run;
    Costs cst;
    double *K;                 // (B,N,N+1,NU,NX)
    double *beta, *beta_f;     // (B,N,N,NI) (B,N+1,NIF)
    double eps;
};

template <int NX, int NU>
__host__ __device__ constexpr int sweep_lds_doubles() {
    // sA sS sYm sAcl sSn sPhi sPhi2 (7 NX^2, NW == NX) + sB sX sF sK sPu (5 NX NU) + sH (NU^2) + sC (NX+NU)
    return 7 * NX * NX + 5 * NX * NU + NU * NU + (NX + NU) + 8;
}

template <int NX, int NU>
__global__ __launch_bounds__(64) void k_sweep(SweepArgs a) {
    using L = Lay<NX, NU>;
    constexpr int NZ = L::NZ, NI = L::NI, NIF = L::NIF, NW = NX;
    const int N = a.N, lane = threadIdx.x;
    // XCD-aware mapping: blocks i and i+8 share an XCD; keep all columns of an instance on one XCD so its
    // A_k/B_k stay in that XCD's L2 (speed only; any mapping is correct).
    const int ncol = N + 1;
    int b, j;
    {
        const int bid = blockIdx.x;
        const int Bfull = (a.B / 8) * 8;
        if (bid < Bfull * ncol) {
            const int xcd = bid % 8, slot = bid / 8;
            b = (slot / ncol) * 8 + xcd; j = slot % ncol;
        } else {
            const int r = bid - Bfull * ncol;
            b = Bfull + r / ncol; j = r % ncol;
        }
    }
    if (b >= a.B) return;
    if (a.run && !a.run[b]) return;
    extern __shared__ double sm[];
    double *p = sm;
    double *sA = p; p += NX * NX; double *sS = p; p += NX * NX; double *sYm = p; p += NX * NX; double *sAcl = p; p += NX * NX;
    double *sSn = p; p += NX * NX; double *sPhi = p; p += NX * NW; double *sPhi2 = p; p += NX * NW;
    double *sB = p; p += NX * NU; double *sX = p; p += NX * NU; double *sF = p; p += NX * NU; double *sK = p; p += NX * NU;
    double *sPu = p; p += NU * NW; double *sH = p; p += NU * NU; double *sC = p; p += NZ;
    const double *gA = a.A + (size_t)b * N * NX * NX, *gB = a.Bm + (size_t)b * N * NX * NU;
    const double *eta = a.eta + (size_t)b * N * N * NI, *eta_f = a.eta_f + (size_t)b * (N + 1) * NIF;
    double *gK = a.K + (size_t)b * N * (N + 1) * NU * NX;
    double *beta = a.beta + (size_t)b * N * N * NI, *beta_f = a.beta_f + (size_t)b * (N + 1) * NIF;

    // terminal: S[N,j] = Gf' diag(eta_f[j]) Gf + Q_reg_f   (Gf = [I;-I], Q_reg_f diagonal)
#pragma unroll
    for (int o = lane; o < NX * NX; o += 64) {
        const int i = o / NX, jj = o % NX;
        sS[o] = (i == jj) ? eta_f[j * NIF + i] + eta_f[j * NIF + NX + i] + a.cst.Qregfd[i] : 0.0;
    }
    wla::wsync();
    for (int k = N - 1; k >= j; k--) {
        const double *Ak = gA + (size_t)k * NX * NX, *Bk = gB + (size_t)k * NX * NU;
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) sA[o] = Ak[o];
#pragma unroll
        for (int o = lane; o < NX * NU; o += 64) sB[o] = Bk[o];
        const double *e = eta + ((size_t)k * N + j) * NI;
        if (lane < NZ) sC[lane] = e[lane] + e[NZ + lane] + (lane < NX ? a.cst.Qregd[lane] : a.cst.Rregd[lane - NX]);
        wla::wsync();
        wla::gemm<NU, NX, NX, true, false>(sB, NU, sS, NX, sX, NX, 1.0, 0.0, lane);   // x = B' S   (NU x NX)
        wla::gemm<NX, NX, NX, true, false>(sA, NX, sS, NX, sYm, NX, 1.0, 0.0, lane);  // y = A' S   (NX x NX)
        wla::wsync();
        wla::gemm<NU, NU, NX, false, false>(sX, NX, sB, NU, sH, NU, 1.0, 0.0, lane);  // H = x B
        wla::gemm<NU, NX, NX, false, false>(sX, NX, sA, NX, sF, NX, 1.0, 0.0, lane);  // F = x A
        wla::wsync();
        if (lane < NU) sH[lane * NU + lane] += sC[NX + lane];
        wla::wsync();
        // K = -H^{-1} F, one column per lane
        if (lane < NX) {
            double f[NU];
#pragma unroll
            for (int u = 0; u < NU; u++) f[u] = sF[u * NX + lane];
            wla::spd_solve_small<NU>(sH, NU, f);
#pragma unroll
            for (int u = 0; u < NU; u++) sK[u * NX + lane] = -f[u];
        }
        wla::wsync();
        double *Kg = gK + ((size_t)k * (N + 1) + j) * NU * NX;
#pragma unroll
        for (int o = lane; o < NU * NX; o += 64) Kg[o] = sK[o];
        // Acl = A + B K
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) {
            const int i = o / NX, jj = o % NX;
            double s = sA[o];
#pragma unroll
            for (int u = 0; u < NU; u++) s = fma(sB[i * NU + u], sK[u * NX + jj], s);
            sAcl[o] = s;
        }
        wla::wsync();
        wla::gemm<NX, NX, NX, false, false>(sYm, NX, sAcl, NX, sSn, NX, 1.0, 0.0, lane);  // y (A + B K)
        wla::wsync();
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) {
            const int i = o / NX, jj = o % NX;
            sS[o] = 0.5 * (sSn[o] + sSn[jj * NX + i]) + ((i == jj) ? sC[i] : 0.0);
        }
        wla::wsync();
    }
    // propagate column j and accumulate row norms
    const double *Eg = a.E + (a.E_per_instance ? (size_t)b * (N + 1) * NX * NW : 0) + (size_t)j * NX * NW;
#pragma unroll
    for (int o = lane; o < NX * NW; o += 64) sPhi[o] = Eg[o];
    wla::wsync();
    double *Pc = sPhi, *Pn = sPhi2;
    for (int k = j; k < N; k++) {
        const double *Ak = gA + (size_t)k * NX * NX, *Bk = gB + (size_t)k * NX * NU;
        const double *Kg = gK + ((size_t)k * (N + 1) + j) * NU * NX;
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) sA[o] = Ak[o];
#pragma unroll
        for (int o = lane; o < NX * NU; o += 64) { sB[o] = Bk[o]; sK[o] = Kg[o]; }
        wla::wsync();
        wla::gemm<NU, NW, NX, false, false>(sK, NX, Pc, NW, sPu, NW, 1.0, 0.0, lane);  // Phi_u = K Phi_x
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) {
            const int i = o / NX, jj = o % NX;
            double s = sA[o];
#pragma unroll
            for (int u = 0; u < NU; u++) s = fma(sB[i * NU + u], sK[u * NX + jj], s);
            sAcl[o] = s;
        }
        wla::wsync();
        // beta[k,j,i] = max(|| row i of [Phi_x;Phi_u] ||^2, eps), rows i and NZ+i of G=[I;-I] coincide
        if (lane < NZ) {
            const double *row = (lane < NX) ? Pc + lane * NW : sPu + (lane - NX) * NW;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w++) s = fma(row[w], row[w], s);
            s = fmax(s, a.eps);
            double *bo = beta + ((size_t)k * N + j) * NI;
            bo[lane] = s; bo[NZ + lane] = s;
        }
        wla::gemm<NX, NW, NX, false, false>(sAcl, NX, Pc, NW, Pn, NW, 1.0, 0.0, lane);
        wla::wsync();
        double *t = Pc; Pc = Pn; Pn = t;
    }
    if (lane < NX) {
        const double *row = Pc + lane * NW;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w++) s = fma(row[w], row[w], s);
        s = fmax(s, a.eps);
        beta_f[j * NIF + lane] = s; beta_f[j * NIF + NX + lane] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// backoff sums + tightened bounds (fast_SLS_jit.py:173-186, 556-569) ; one workgroup per instance
// ------------------------------------------------------------------------------------------------
struct TightenArgs {
    int B, N, NX, NU;
    const double *beta, *beta_f, *g, *gf_raw, *c;
    const int *run;
    double *backoff, *backoff_f, *backoff_x, *backoff_u, *ubg;
    int write_ubg;
};
__global__ void k_tighten(TightenArgs a) {
    const int b = blockIdx.x;
    if (a.run && !a.run[b]) return;
    const int NX = a.NX, NU = a.NU, NZ = NX + NU, NI = 2 * NZ, NIF = 2 * NX, N = a.N, SR = NX + NI, mb = N * SR + NIF;
    const double *be = a.beta + (size_t)b * N * N * NI, *bf = a.beta_f + (size_t)b * (N + 1) * NIF;
    double *bo = a.backoff + (size_t)b * N * NI, *bof = a.backoff_f + (size_t)b * NIF;
    double *bx = a.backoff_x + (size_t)b * (N + 1) * NX, *bu = a.backoff_u + (size_t)b * N * NU;
    double *ub = a.ubg + (size_t)b * mb;
    for (int o = threadIdx.x; o < N * NI; o += blockDim.x) {
        const int k = o / NI, i = o % NI;
        double acc = 0.0;
        for (int j = 0; j <= k; j++) acc += sqrt(be[((size_t)k * N + j) * NI + i]);
        bo[o] = acc;
        if (i < NX) bx[k * NX + i] = acc;
        else if (i < NZ) bu[k * NU + (i - NX)] = acc;
        if (a.write_ubg) ub[k * SR + NX + i] = a.g[((size_t)b * N + k) * NI + i] - acc;   // no +eps (quirk q3)
    }
    for (int o = threadIdx.x; o < NIF; o += blockDim.x) {
        double acc = 0.0;
        for (int j = 0; j <= N; j++) acc += sqrt(bf[j * NIF + o]);
        bof[o] = acc;
        if (o < NX) bx[N * NX + o] = acc;
        if (a.write_ubg) ub[N * SR + o] = a.gf_raw[o] - acc;                               // raw gf (quirk q2)
    }
    if (a.write_ubg)
        for (int o = threadIdx.x; o < N * NX; o += blockDim.x) ub[(o / NX) * SR + (o % NX)] = -a.c[(size_t)b * N * NX + o];
}

// initialize_backoff (fast_SLS_jit.py:444-454)
struct InitBackoffArgs { int B, N, NX, NU; double eps; const int *run; double *beta, *beta_f, *backoff, *backoff_f, *backoff_x, *backoff_u; };
__global__ void k_init_backoff(InitBackoffArgs a) {
    const int b = blockIdx.x;
    if (a.run && !a.run[b]) return;
    const int NZ = a.NX + a.NU, NI = 2 * NZ, NIF = 2 * a.NX, N = a.N;
    const double sq = sqrt(a.eps);
    for (int o = threadIdx.x; o < N * N * NI; o += blockDim.x) a.beta[(size_t)b * N * N * NI + o] = a.eps;
    for (int o = threadIdx.x; o < (N + 1) * NIF; o += blockDim.x) a.beta_f[(size_t)b * (N + 1) * NIF + o] = a.eps;
    for (int o = threadIdx.x; o < N * NI; o += blockDim.x) a.backoff[(size_t)b * N * NI + o] = N * sq;
    for (int o = threadIdx.x; o < NIF; o += blockDim.x) a.backoff_f[(size_t)b * NIF + o] = (N + 1) * sq;
    for (int o = threadIdx.x; o < (N + 1) * a.NX; o += blockDim.x) a.backoff_x[(size_t)b * (N + 1) * a.NX + o] = 0.0;
    for (int o = threadIdx.x; o < N * a.NU; o += blockDim.x) a.backoff_u[(size_t)b * N * a.NU + o] = 0.0;
}

}  // namespace slsqp
