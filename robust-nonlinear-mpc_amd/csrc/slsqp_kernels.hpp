// slsqp_kernels.hpp -- device kernels of the batched fast-SLS QP path for gfx950 (MI355X).
//
// Mapping: ONE 64-lane wavefront per MPC instance (QP kernel, shared Riccati recursion) or per (instance, disturbance column)
// (propagation / general sweep kernels); workgroups are a single wave, so thousands of independent horizon recursions are in
// flight and the scheduler interleaves them on every SIMD.  Stage blocks (A_k, B_k, stored inverses D_k^-1) are staged
// HBM/L2 -> registers -> LDS with coalesced loads; the small dense algebra runs out of LDS / registers (wave_la.hpp).
//
// Reference behaviour restated here (citations relative to antoineleeman/robust-nonlinear-mpc):
//   k_qp_solve                the QP of QP.solve            solver/qp_jit.py:362-402 (data layout :77-192)
//   k_sweep_ric1 / _prop      _backward_solve_numba / _propagate / _backoff_from_phi (beta part), first fast-SLS iteration   fast_SLS_jit.py:43-158
//   k_sweep / k_sweep_gen     the same for later iterations / a general constraint matrix G
//   k_tighten                 _backoff_from_phi (sums) + update_tightening tail             :160-188, :556-569
//   k_set_bounds              QP.update_dynamics + offset_constraints  solver/qp_jit.py:268-273, 595-610
//   k_lin_val / _tan / _vec   SCP_SLS.update_jacobian       solver/SCP_SLS_jit.py:251-366
//   k_cl_*                    socp_step, reset_warm_start, plant step of the closed-loop scripts   SCP_SLS_jit.py:404-551
//   (evaluate_dual_eta :475-487 and check_convergence_socp :581-600 run inside k_after_qp, slsqp_api.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "dynamics.hpp"
#include "wave_la.hpp"

namespace slsqp {

constexpr double BIGB = 1e19;   // |bound| above this is "infinite" (the reference maps +-inf to +-1e20, qp_jit.py:382)
constexpr double EPS_PIN = 1e-10;
constexpr int ST_INIT = -1;

// NE_MFMA (default 1): the factor sweep's products T = M1 Dinv and D_k = M1 A' + B diag(pi) B' - T M1' + diag run on the fp64 matrix core
// for NX >= 13 (wla::gemm_mfma / build_Y_mfma); 0 = vector-ALU versions.
#ifndef NE_MFMA
#define NE_MFMA 1
#endif
// NE_GJ_MFMA (default 1): the inverse of D_k by the matrix-core Gauss-Jordan sweep (wla::spd_inv_gj_mfma); 0 = the 2x2-block vector-ALU sweep
#ifndef NE_GJ_MFMA
#define NE_GJ_MFMA 1
#endif

struct Costs {  // batch-constant diagonal weights (device pointers)
    const double *Qd, *Rd, *Qfd;        // diag of Q, R, Qf (P = 2*blkdiag)
    const double *Qregd, *Rregd, *Qregfd;
    double prox;                        // added to every diagonal entry of blkdiag(Q,R,..,Qf) (trust-region weight of the nominal initialiser; 0 on the path)
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
// QP kernel: active-set iteration on the block-tridiagonal normal equations first, primal-dual interior point as the fall-back
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
    double *ws;               // scratch (B, qp_ws_doubles(n,N,NX)): the IPM's n-vectors
    double *state;            // scratch (B,40): QpState
    double *primal;           // (B,n)
    double *dual;             // (B,mb)
    double *cost;             // (B)
    double *pin_dual;         // (B,NX) multipliers of the x0-pin rows (may be NULL)
    double *kkt;              // (B,8) [0..3] accepted solution: stationarity, box violation, multiplier-sign violation, mu;
                              //       [4],[5] last polish attempt: stationarity, box; [6],[7] factor sweeps / solves used
    int *status, *iters;      // (B)
    int max_iter;
    double eps;
    double init_s, init_lam;  // starting slack floor / multiplier of the interior point
    int warm_rounds;          // active-set correction rounds allowed in a warm attempt before falling back
    int warm;                 // 1: try an active-set polish from the previous solution of each instance first
    const double *prox;       // per-instance proximal weight added to the diagonal cost (element b * prox_stride), or NULL
    int prox_stride;
    unsigned long long *inst_launches;   // [0] += 1 per instance forward sweep, [1] += 1 per factorising one, [2] += stages it factorised, [4] += block solves whose backward sweep was skipped; [3] += QP solves that ran at least one block solve (an instance whose x0 pin
                                         // contradicts its own box is flagged at once and not counted)
                                         // (roofline accounting of bench.py)
    int *qpstat;              // (B,2,8) or NULL: per instance and slot [its, block solves, factorising ones, active inequality rows, started warm,
    int stat_slot;            //   active-set correction rounds, status, fell back to the interior point]; slot = 0 first QP of a fast-SLS call, 1 its last QP
    int snap_take;            // 1: keep a copy of this solve's interior-point iterate once mu <= snap_mu * max(1,|q|inf) (start for the next QP of the call)
    int snap_use;             // 1: an instance whose warm active-set attempt fails restarts its interior point from that copy (same A, B, q; other bounds)
    double snap_mu;
    double call_id;           // identifies the fast-SLS call (a copy is only valid within the call that took it: same A, B, q)
    const double *call_ids;   // (B) or NULL: per-instance call ids instead (slsqp_cl_run: instances of one launch may be in different MPC steps)
    int as_first;             // 1: a cold solve first tries the active-set iteration from the empty set (the equality-constrained optimum of P_INIT)
    int as_rounds;            // rounds such an attempt may take (a warm one: warm_rounds)
    int *diag;                // QP_DIAG_SPAN builds: (B,2,16,2) first / last stage whose set entry changed, per active-set round
    const int *shift_stepno;  // NULL, or (B): warm_shift applies to the instances with shift_stepno[b] > 0 only (k_cl_loop: every instance is at its own MPC step)
    int warm_shift;           // 1: the first QP's warm set is the previous call's set moved one stage towards the start of the horizon (receding horizon)
    int as_warm_max_set;      // a QP that follows another one of the same call (tightened bounds) skips the warm attempt and goes straight to the
                              // interior point when the set it would start from has more entries than this (0 = never): from ~30 active bounds a
                              // tightening moves so many of them that the rounds cost more than the interior point
    int as_warm_last;         // the LAST QP of a call (stat_slot 1: the tightened one) may start from the certified set of the previous call's last QP, moved with
                              // the horizon like the first QP's (warm_shift): 1 = when the first QP's set is too large for as_warm_max_set (or gave no
                              // certified set), i.e. where the interior point would run otherwise; 2 = whenever such a set exists; 0 = never
    double pinf_eps;          // primal-infeasibility certificate of the interior point (OSQP's eps_prim_inf, default 1e-4; 0 = off): see start_iter
    int as_max_viol;          // an attempt is abandoned when a solve leaves more violated bounds than this, or more than 2 x the previous round's + 8
    int n_refine;             // refinement solves per polish (1 in fp64, 3 with fp32 factorisations)
    double early_ctol;        // tolerance (relative to max(1,|q|inf)) of the look at the un-refined polish solve: its accuracy class
};

// LDS layout of one QP wave: NX x NX buffers A_k | [T = M1 Dinv_{k-1}, vector-ALU path only] | Dinv_{k-1}, then T, then D_k | M1 -> Dinv_k (the last two rotate),
// B_k, and a handful of stage vectors.  Everything of size n (IPM vectors) lives in an HBM/L2 workspace.
template <int NX, int NU>
struct QpLds {
    static constexpr int NZ = NX + NU, MM = NX * NX;
    static constexpr bool MFMA = (NE_MFMA != 0) && NX >= 13;   // then T and D_k are built in place in the Dinv_{k-1} buffer: 3 NX^2 buffers
    static constexpr int oA = 0, oL1 = MM, oP = MFMA ? MM : 2 * MM, oQ = oP + MM, oB = oQ + MM, oPiS = oB + NX * NU, oVS = oPiS + NZ + NX,
                         oWp = oVS + NZ + NX, oT1 = oWp + NX, oT2 = oT1 + NX, oT3 = oT2 + NX, TOTAL = oT3 + NX + 1;
};
template <int NX, int NU>
__host__ __device__ constexpr int qp_lds_doubles(int) { return QpLds<NX, NU>::TOTAL; }

// workspace arrays per instance (doubles): 12 n-vectors + 3 (N*NX)-vectors of the solver, then the copy of an interior-point iterate
// (6 n-vectors + 1 (N*NX)-vector: z, s_u, s_l, lambda_u, lambda_l, E'nu, nu), then the certified active set of the call's first QP (1 n-vector), u of the last forward sweep
// (1 (N*NX)-vector) and the certified active set of the call's last QP (1 n-vector)
__host__ __device__ inline size_t qp_ws_doubles(int n, int N, int NX) { return (size_t)20 * n + (size_t)5 * N * NX; }

template <int NX, int NU>
struct NeG {   // global-memory operands of the sweeps (this instance)
    const double *A, *Bm, *ub, *lb;
    double *Linv, *PI, *V, *G, *W;
    double *UF;      // u_k of the last forward sweep (fp64 kernels; the backward sweep turns them into nu_k in W)
    int N;
};

// Stage blocks in flight between global memory and LDS.  A_k and B_k travel as pairs of doubles (one 16-byte load and one ds_write2_b64 per pair:
// 3 + 1 loads per lane for the rocket instead of 5 + 2); the stored inverse D_k^-1 travels as the 2x2 blocks of its lower triangle in the lane
// layout of the Gauss-Jordan sweep that produced it (wla::spd_inv_gj: 4 doubles per lane on 45 lanes = 1440 B instead of the full symmetric
// 2312 B) and is mirrored to the full matrix on its way into LDS.
template <int NX, int NU>
struct StageIO {
    static constexpr int MM = NX * NX, NB = NX * NU, PA = (MM + 1) / 2, PB = (NB + 1) / 2, RA = (PA + 63) / 64, RB = (PB + 63) / 64;
    static constexpr int NT = wla::gj_blocks<NX>(), DSTR = 4 * NT;      // doubles per stored inverse
    static_assert(NT <= 64 && DSTR <= MM, "block-packed inverse fits the scratch of the full one");
    wla::d2 A[RA], B[RB], L[2];
    __device__ __forceinline__ void load_AB(const double *Ak, const double *Bk, int lane) {
#pragma unroll
        for (int r = 0; r < RA; r++) A[r] = wla::ld2(Ak + 2 * min(r * 64 + lane, PA - 1));      // (an odd block's last pair reads one double of its successor: never stored)
#pragma unroll
        for (int r = 0; r < RB; r++) B[r] = wla::ld2(Bk + 2 * min(r * 64 + lane, PB - 1));
    }
    __device__ __forceinline__ void load_L(const double *Lk, int lane) {
        const int l = min(lane, NT - 1);
        L[0] = wla::ld2(Lk + 4 * l); L[1] = wla::ld2(Lk + 4 * l + 2);
    }
    __device__ __forceinline__ void store_AB(double *sA, double *sB, int lane) const {
#pragma unroll
        for (int r = 0; r < RA; r++) {
            const int p = r * 64 + lane;
            if (2 * p + 1 < MM) wla::st2(sA + 2 * p, A[r]);
            else if (2 * p < MM) sA[2 * p] = A[r].x;
        }
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const int p = r * 64 + lane;
            if (2 * p + 1 < NB) wla::st2(sB + 2 * p, B[r]);
            else if (2 * p < NB) sB[2 * p] = B[r].x;
        }
    }
    __device__ __forceinline__ void store_L(double *sL, int lane) const { wla::gj_blocks_to_lds<NX>(L[0].x, L[0].y, L[1].x, L[1].y, sL, NX, lane); }
};

// Forward sweep over the horizon of the block-tridiagonal normal equations  Y nu = b,  Y = E Pi E':
//   optional (re)factorisation, block LDL':  D_k = Y_kk - O_k D_{k-1}^-1 O_k',  O_k = Y_{k,k-1} = -A_k diag(pi_x,k)
//   (explicit symmetric inverses Dinv_k kept, written block-packed to HBM scratch),
//   rhs b_k = E_k v - eflag * e_k, forward elimination t_k = b_k - O_k u_{k-1},  u_k = Dinv_k t_k  (u_k stored in UF).
template <int NX, int NU>
__device__ __forceinline__ int ne_forward(double *sm, const NeG<NX, NU> g, bool factor_all, double eflag, double delta, int lane,
                                          double *bmax_out = nullptr, int k0 = 0, int ks = 0) {
    // k0 > 0 (active-set rounds): the stages before k0 keep the factorisation of the previous round -- their D_k depend on Pi of the stages
    // <= k + 1 only, and no entry of Pi changed there -- so their inverses are read back from the scratch like in a solve-only sweep.
    // ks > 0 (<= k0): the sweep starts at stage ks.  The right-hand side of an active-set round differs from the previous round's only where the
    // set changed (v = -pi q on free elements, the bound on fixed ones), i.e. from stage k0 on, so u_0 .. u_{ks-1} of the previous sweep (kept in
    // UF) still hold and only u_{ks-1} and Dinv_{ks-1} are read back.
    double bmax = 0.0;
    using Ld = QpLds<NX, NU>;
    using IO = StageIO<NX, NU>;
    constexpr int NZ = NX + NU, SR = NX + 2 * NZ, MM = NX * NX;
    double *sA = sm + Ld::oA, *sL1 = sm + Ld::oL1, *sB = sm + Ld::oB, *sPiS = sm + Ld::oPiS, *sVS = sm + Ld::oVS;
    double *sWp = sm + Ld::oWp, *sT1 = sm + Ld::oT1, *sT2 = sm + Ld::oT2;
    double *Lprev = sm + Ld::oP, *Lcur = sm + Ld::oQ;
    int fail = 0;
    // software pipeline: stage k+1's blocks are fetched HBM/L2 -> registers while stage k is being processed
    IO io;
    double rPi = 0.0, rV = 0.0, rE = 0.0;
    auto prefetch = [&](int k) {
        io.load_AB(g.A + (size_t)k * MM, g.Bm + (size_t)k * NX * NU, lane);
        if (!(factor_all && k >= k0)) io.load_L(g.Linv + (size_t)k * IO::DSTR, lane);
        const int ls = min(lane, NZ + NX - 1), lx = min(lane, NX - 1);
        rPi = g.PI[k * NZ + ls]; rV = g.V[k * NZ + ls];
        if (eflag != 0.0) rE = 0.5 * (g.ub[k * SR + lx] + g.lb[k * SR + lx]);
    };
    if (ks > 0) {
        io.load_L(g.Linv + (size_t)(ks - 1) * IO::DSTR, lane);
        io.store_L(Lprev, lane);
        if (lane < NX) sWp[lane] = g.UF[(ks - 1) * NX + lane];
    }
    prefetch(ks);
    for (int k = ks; k < g.N; k++) {
        const bool factor = factor_all && k >= k0;
        io.store_AB(sA, sB, lane);
        if (!factor) io.store_L(Lcur, lane);
        if (lane < NZ + NX) { sPiS[lane] = rPi; sVS[lane] = rV; }
        const double ek = rE;
        wla::wsync();
        if (k + 1 < g.N) prefetch(k + 1);
        if (factor) {
            // M1 = A diag(pi_x,k) (into Lcur's buffer, dead until the inverse is written); the MFMA path scales its operands on the fly
            if constexpr (!Ld::MFMA) {
#pragma unroll
                for (int o = lane; o < MM; o += 64) Lcur[o] = (k > 0) ? sA[o] * sPiS[o % NX] : 0.0;
                wla::wsync();
            }
            if (k > 0) {   // T = M1 Dinv_{k-1}   (Dinv symmetric, so the NT product is the NN one)
                if constexpr (Ld::MFMA) wla::gemm_mfma<NX, NX, NX, false, false, false, true>(sA, NX, Lprev, NX, Lprev, NX, lane, nullptr, 0, sPiS);   // in place: T replaces Dinv_{k-1}
                else wla::gemm_nt_blk<NX, NX, NX, 3, 2>(Lcur, NX, Lprev, NX, sL1, NX, 1.0, lane);
                wla::wsync();
            }
            // lower(D_k) = M1 A' + B diag(pi_u) B' + diag(pi_x,k+1) + delta - T M1'      (block LDL': D_k = Y_kk - O D_{k-1}^-1 O')
            // D_k is built where Dinv_{k-1} was (dead once T is formed: u_{k-1} is kept in sWp), inverted from there into M1's buffer
            double *sY = Lprev;
            if constexpr (Ld::MFMA) wla::build_Y_mfma<NX, NU>(sA, sPiS, sB, sPiS + NX, Lprev, k > 0, sPiS + NZ, delta, sY, lane);   // in place over T
            else wla::build_Y_lower<NX, NU>(Lcur, sA, sB, sPiS + NX, sL1, k > 0, sPiS + NZ, delta, sY, lane);
            wla::wsync();
            // the inverse goes to LDS (full, for this stage's and the next stage's products) and, block-packed from the registers, to the scratch
            if constexpr (Ld::MFMA && NE_GJ_MFMA != 0) fail |= wla::spd_inv_gj_mfma<NX>(sY, NX, Lcur, NX, g.Linv + (size_t)k * IO::DSTR, lane);      // one rank-2 MFMA update per 2x2 pivot
            else fail |= wla::spd_inv_gj<NX>(sY, NX, Lcur, NX, g.Linv + (size_t)k * IO::DSTR, lane);
        }
        // rhs b = A v_x + B v_u - v_x,k+1 - eflag e_k  (+ A (pi_x,k .* Linv_{k-1}' w_{k-1}))
        // t_k = A (v_x + pi_x,k .* u_{k-1}) + B v_u - v_x,k+1 - eflag e_k   (u_{k-1} = Dinv_{k-1} t_{k-1} kept in sWp; the second term is -O_k u_{k-1})
        if (lane < NX) sT1[lane] = sVS[lane] + ((k > 0) ? sWp[lane] * sPiS[lane] : 0.0);
        wla::wsync();
        double b = wla::matvec_split3<NX, NX, false>(sA, NX, sT1, lane) + wla::matvec_split3<NX, NU, false>(sB, NU, sVS + NX, lane);
        if (lane < NX) b += -sVS[NZ + lane] - eflag * ek; else b = 0.0;
        if (lane < NX) sT2[lane] = b;
        bmax = fmax(bmax, fabs(b));
        wla::wsync();
        const double w = wla::matvec_split3<NX, NX, false>(Lcur, NX, sT2, lane);   // u_k = Dinv_k t_k
        wla::wsync();
        if (lane < NX) { sWp[lane] = w; g.UF[k * NX + lane] = w; }
        double *t = Lcur; Lcur = Lprev; Lprev = t;
        wla::wsync();
    }
    if (bmax_out) *bmax_out = wla::wave_max(bmax);
    return fail;
}

// Backward sweep: nu_k = u_k - Dinv_k O_{k+1}' nu_{k+1} (overwrites W) and G = E' nu.
template <int NX, int NU>
__device__ __forceinline__ void ne_backward(double *sm, const NeG<NX, NU> g, int lane) {
    using Ld = QpLds<NX, NU>;
    using IO = StageIO<NX, NU>;
    constexpr int NZ = NX + NU, MM = NX * NX;
    double *sA = sm + Ld::oA, *sLa = sm + Ld::oQ, *sB = sm + Ld::oB, *sPiS = sm + Ld::oPiS, *sWp = sm + Ld::oWp;
    double *sT1 = sm + Ld::oT1, *sT2 = sm + Ld::oT2, *sT3 = sm + Ld::oT3;
    if (lane < NX) sT3[lane] = 0.0;  // A_{k+1}' nu_{k+1}
    wla::wsync();
    // the sweep is bound by the latency of the stage loads (A_k, B_k, Dinv_k stream through once, 2 kflop per stage): two stages are
    // kept in flight in two register sets
    struct StageRegs { IO io; double Pi, W; };
    StageRegs r0, r1;
    auto load = [&](int k, StageRegs &r) {
        r.io.load_AB(g.A + (size_t)k * MM, g.Bm + (size_t)k * NX * NU, lane);
        r.io.load_L(g.Linv + (size_t)k * IO::DSTR, lane);
        const int lx = min(lane, NX - 1);
        r.Pi = g.PI[(k + 1) * NZ + lx]; r.W = g.UF[k * NX + lx];
    };
    auto stage = [&](int k, StageRegs &r) {
        r.io.store_AB(sA, sB, lane);
        r.io.store_L(sLa, lane);
        if (lane < NX) { sPiS[lane] = r.Pi; sWp[lane] = r.W; }
        wla::wsync();
        if (k >= 2) load(k - 2, r);
        if (lane < NX) sT1[lane] = sPiS[lane] * sT3[lane];
        wla::wsync();
        // nu_k = u_k - Dinv_k O_{k+1}' nu_{k+1} = u_k + Dinv_k (pi_x,k+1 .* A_{k+1}' nu_{k+1})
        const double nu = sWp[lane < NX ? lane : 0] + wla::matvec_split3<NX, NX, false>(sLa, NX, sT1, lane);
        wla::wsync();
        if (lane < NX) {
            g.W[k * NX + lane] = nu;
            g.G[(k + 1) * NZ + lane] = sT3[lane] - nu;
            sT2[lane] = nu;
        }
        wla::wsync();
        const double ga = wla::matvec_split3<NX, NX, true>(sA, NX, sT2, lane);
        const double gb = wla::matvec_split3<NU, NX, true>(sB, NU, sT2, lane);
        wla::wsync();
        if (lane < NX) sT3[lane] = ga;
        if (lane < NU) g.G[k * NZ + NX + lane] = gb;
        wla::wsync();
    };
    load(g.N - 1, r0);
    if (g.N >= 2) load(g.N - 2, r1);
    for (int k = g.N - 1; k >= 0; k -= 2) {
        stage(k, r0);
        if (k >= 1) stage(k - 1, r1);
    }
    if (lane < NX) g.G[lane] = sT3[lane];
}

// ------------------------------------------------------------------------------------------------
// Mixed-precision variant of the two sweeps (opts.precision = 1; BASELINE config 3 "fp32 vs fp64"):
//   fp32: the factorisation (M1, T = M1 Dinv, D_k, Gauss-Jordan inverse), the stored inverses Dinv_k (half the scratch traffic)
//         and the substitutions with them;
//   fp64: everything that defines the equations -- the right-hand sides E v - e, the off-diagonal products A (pi .* u), and
//         G = E' nu -- so the residuals the interior point and the polish see are exact for the iterate they hold.
// The solves are then fp32-accurate Newton / refinement steps on fp64 residuals: classical mixed-precision iterative refinement.
// The polish spends one more refinement solve (3 instead of 2) and ends on the same fp64 KKT certificate; what does not certify
// is solved again by the fp64 kernels (launch_qp).
// ------------------------------------------------------------------------------------------------
template <int NX, int NU>
struct QpLdsMx {   // doubles first (offsets in doubles), then floats (offsets in floats from the float base)
    static constexpr int NZ = NX + NU, MM = NX * NX;
    static constexpr int dA = 0, dB = dA + MM, dPiS = dB + NX * NU, dVS = dPiS + NZ + NX, dWp = dVS + NZ + NX, dT1 = dWp + NX, dT2 = dT1 + NX,
                         dT3 = dT2 + NX, DTOT = dT3 + NX + 1;
    static constexpr int fA = 0, fM = MM, fP = 2 * MM, fY = 3 * MM, fL1 = 4 * MM, fB = 5 * MM, fPi = fB + NX * NU, fT = fPi + NZ + NX, FTOT = fT + NX + 2;
    static constexpr size_t BYTES = 8 * (size_t)DTOT + 4 * (size_t)FTOT;
};

template <int NX, int NU>
__device__ __forceinline__ int ne_forward_mx(double *smd, const NeG<NX, NU> g, bool factor_all, double eflag, double delta, int lane, double *bmax_out, int k0 = 0) {
    double bmax = 0.0;
    using Ld = QpLdsMx<NX, NU>;
    constexpr int NZ = NX + NU, SR = NX + 2 * NZ, MM = NX * NX;
    double *sA = smd + Ld::dA, *sB = smd + Ld::dB, *sPiS = smd + Ld::dPiS, *sVS = smd + Ld::dVS, *sWp = smd + Ld::dWp, *sT1 = smd + Ld::dT1;
    float *smf = (float *)(smd + Ld::DTOT);
    float *fA = smf + Ld::fA, *Lcur = smf + Ld::fM, *Lprev = smf + Ld::fP, *fY = smf + Ld::fY, *fL1 = smf + Ld::fL1, *fB = smf + Ld::fB, *fPi = smf + Ld::fPi,
          *fT = smf + Ld::fT;
    float *Linv_g = (float *)g.Linv;
    int fail = 0;
    constexpr int RA = (MM + 63) / 64, RB = (NX * NU + 63) / 64;
    double rA[RA], rB[RB], rPi = 0.0, rV = 0.0, rE = 0.0;
    float rL[RA];
    auto prefetch = [&](int k) {
        const double *Ak = g.A + (size_t)k * MM, *Bk = g.Bm + (size_t)k * NX * NU;
        const float *Lg = Linv_g + (size_t)k * MM;
#pragma unroll
        for (int r = 0; r < RA; r++) rA[r] = Ak[min(r * 64 + lane, MM - 1)];
#pragma unroll
        for (int r = 0; r < RB; r++) rB[r] = Bk[min(r * 64 + lane, NX * NU - 1)];
        if (!(factor_all && k >= k0)) {
#pragma unroll
            for (int r = 0; r < RA; r++) rL[r] = Lg[min(r * 64 + lane, MM - 1)];
        }
        const int ls = min(lane, NZ + NX - 1), lx = min(lane, NX - 1);
        rPi = g.PI[k * NZ + ls]; rV = g.V[k * NZ + ls];
        if (eflag != 0.0) rE = 0.5 * (g.ub[k * SR + lx] + g.lb[k * SR + lx]);
    };
    prefetch(0);
    for (int k = 0; k < g.N; k++) {
        const bool factor = factor_all && k >= k0;
#pragma unroll
        for (int r = 0; r < RA; r++) { const int o = r * 64 + lane; if (o < MM) { sA[o] = rA[r]; if (factor) fA[o] = (float)rA[r]; else Lcur[o] = rL[r]; } }
#pragma unroll
        for (int r = 0; r < RB; r++) { const int o = r * 64 + lane; if (o < NX * NU) { sB[o] = rB[r]; if (factor) fB[o] = (float)rB[r]; } }
        if (lane < NZ + NX) { sPiS[lane] = rPi; sVS[lane] = rV; fPi[lane] = (float)rPi; }
        const double ek = rE;
        wla::wsync();
        if (k + 1 < g.N) prefetch(k + 1);
        if (factor) {
#pragma unroll
            for (int o = lane; o < MM; o += 64) Lcur[o] = (k > 0) ? fA[o] * fPi[o % NX] : 0.0f;   // M1 = A diag(pi_x,k)
            wla::wsync();
            if (k > 0) {
                wla::gemm_nt_blk<NX, NX, NX, 3, 2>(Lcur, NX, Lprev, NX, fL1, NX, 1.0f, lane);     // T = M1 Dinv_{k-1}
                wla::wsync();
            }
            wla::build_Y_lower<NX, NU>(Lcur, fA, fB, fPi + NX, fL1, k > 0, fPi + NZ, (float)delta, fY, lane);
            wla::wsync();
            fail |= wla::spd_inv_gj<NX>(fY, NX, Lcur, NX, (float *)nullptr, lane);
            float *Lg = Linv_g + (size_t)k * MM;
#pragma unroll
            for (int o = lane; o < MM; o += 64) Lg[o] = Lcur[o];
        }
        // right-hand side in fp64: b = A v_x + B v_u - v_x,k+1 - eflag e_k  (+ A (pi_x,k .* u_{k-1}))
        double b = 0.0;
        if (lane < NX) {
            b = -sVS[NZ + lane] - eflag * ek;
#pragma unroll
            for (int m = 0; m < NX; m++) b = fma(sA[lane * NX + m], sVS[m], b);
#pragma unroll
            for (int m = 0; m < NU; m++) b = fma(sB[lane * NU + m], sVS[NX + m], b);
        }
        if (k > 0) {
            if (lane < NX) sT1[lane] = sWp[lane] * sPiS[lane];
            wla::wsync();
            if (lane < NX) {
#pragma unroll
                for (int m = 0; m < NX; m++) b = fma(sA[lane * NX + m], sT1[m], b);
            }
        }
        if (lane < NX) fT[lane] = (float)b;
        bmax = fmax(bmax, fabs(b));
        wla::wsync();
        const double w = (double)wla::matvec_split3<NX, NX, false>(Lcur, NX, fT, lane);   // u_k = Dinv_k t_k  (fp32)
        wla::wsync();
        if (lane < NX) { sWp[lane] = w; g.W[k * NX + lane] = w; }
        float *t = Lcur; Lcur = Lprev; Lprev = t;
        wla::wsync();
    }
    if (bmax_out) *bmax_out = wla::wave_max(bmax);
    return fail;
}

template <int NX, int NU>
__device__ __forceinline__ void ne_backward_mx(double *smd, const NeG<NX, NU> g, int lane) {
    using Ld = QpLdsMx<NX, NU>;
    constexpr int NZ = NX + NU, MM = NX * NX;
    double *sA = smd + Ld::dA, *sB = smd + Ld::dB, *sPiS = smd + Ld::dPiS, *sWp = smd + Ld::dWp, *sT2 = smd + Ld::dT2, *sT3 = smd + Ld::dT3;
    float *smf = (float *)(smd + Ld::DTOT);
    float *sLa = smf + Ld::fM, *fT = smf + Ld::fT;
    const float *Linv_g = (const float *)g.Linv;
    if (lane < NX) sT3[lane] = 0.0;
    wla::wsync();
    constexpr int RA = (MM + 63) / 64, RB = (NX * NU + 63) / 64;
    double rA[RA], rB[RB], rPi = 0.0, rW = 0.0;
    float rL[RA];
    auto prefetch = [&](int k) {
        const double *Ak = g.A + (size_t)k * MM, *Bk = g.Bm + (size_t)k * NX * NU;
        const float *Lg = Linv_g + (size_t)k * MM;
#pragma unroll
        for (int r = 0; r < RA; r++) { const int o = min(r * 64 + lane, MM - 1); rA[r] = Ak[o]; rL[r] = Lg[o]; }
#pragma unroll
        for (int r = 0; r < RB; r++) rB[r] = Bk[min(r * 64 + lane, NX * NU - 1)];
        const int lx = min(lane, NX - 1);
        rPi = g.PI[(k + 1) * NZ + lx]; rW = g.W[k * NX + lx];
    };
    prefetch(g.N - 1);
    for (int k = g.N - 1; k >= 0; k--) {
#pragma unroll
        for (int r = 0; r < RA; r++) { const int o = r * 64 + lane; if (o < MM) { sA[o] = rA[r]; sLa[o] = rL[r]; } }
#pragma unroll
        for (int r = 0; r < RB; r++) { const int o = r * 64 + lane; if (o < NX * NU) sB[o] = rB[r]; }
        if (lane < NX) { sPiS[lane] = rPi; sWp[lane] = rW; }
        wla::wsync();
        if (k > 0) prefetch(k - 1);
        if (lane < NX) fT[lane] = (float)(sPiS[lane] * sT3[lane]);
        wla::wsync();
        const double nu = sWp[lane < NX ? lane : 0] + (double)wla::matvec_split3<NX, NX, false>(sLa, NX, fT, lane);
        wla::wsync();
        if (lane < NX) {
            g.W[k * NX + lane] = nu;
            g.G[(k + 1) * NZ + lane] = sT3[lane] - nu;
            sT2[lane] = nu;
        }
        wla::wsync();
        const double ga = wla::matvec_split3<NX, NX, true>(sA, NX, sT2, lane);     // E' nu in fp64: consistent with the stored nu
        const double gb = wla::matvec_split3<NU, NX, true>(sB, NU, sT2, lane);
        wla::wsync();
        if (lane < NX) sT3[lane] = ga;
        if (lane < NU) g.G[k * NZ + NX + lane] = gb;
        wla::wsync();
    }
    if (lane < NX) g.G[lane] = sT3[lane];
}

// per-element constants of the stage-ordered primal vector, re-read from L2-resident inputs where needed
struct Elem { double hi, lo, q, pd; bool fu, fl, fr; };
template <int NX, int NU>
__device__ __forceinline__ Elem elem_of(int e, int n, int N, const double *ub, const double *qg, const Costs &cst) {
    constexpr int NZ = NX + NU, SR = NX + 2 * NZ;
    Elem r;
    const int k = e / NZ, i = e % NZ;
    r.q = qg[e];
    if (k < N) {
        r.pd = 2.0 * ((i < NX ? cst.Qd[i] : cst.Rd[i - NX]) + cst.prox);
        r.hi = ub[k * SR + NX + i];
        r.lo = -ub[k * SR + NX + NZ + i];
    } else {
        r.pd = 2.0 * (cst.Qfd[i] + cst.prox);
        r.hi = ub[N * SR + i];
        r.lo = -ub[N * SR + NX + i];
    }
    r.fr = e >= NX;
    r.fu = r.fr && r.hi < BIGB;
    r.fl = r.fr && r.lo > -BIGB;
    return r;
}

// ------------------------------------------------------------------------------------------------
// QP solver = phase machine: per "tick" a forward sweep, a backward sweep and the phase logic below (k_qp_solve loops over them), state per
// instance in HBM.
// Algorithm (per instance):
//   1. equality-constrained optimum (bounds ignored) as starting point,
//   2. Mehrotra predictor-corrector interior point; every Newton system is reduced to the block-tridiagonal
//      normal equations  (E Pi E') dnu = rhs,  Pi = (P + Sigma)^-1 diagonal, factorised stage by stage,
//   3. active-set polish (the OSQP-polish idea, qp_jit.py:546 `polishing=True`): fix the variables the interior
//      point identifies as active, re-solve the KKT system exactly with a refinement step, accept only if the
//      KKT certificate (stationarity, box feasibility, multiplier signs) holds to 1e-9; otherwise correct the
//      active set (primal-dual active-set step) and repeat, at most 6 times.
// One tick = one block-tridiagonal solve (forward sweep, backward sweep) + the elementwise work that consumes it and
// prepares the next right-hand side.  Instances advance independently (different phases coexist in one launch);
// finished instances exit at once.  Splitting the former single kernel removed 340 VGPR + 382 SGPR spills.
// ------------------------------------------------------------------------------------------------
enum { P_INIT = 0, P_PRED = 1, P_CORR = 2, P_POL0 = 3, P_POL1 = 4, P_POL2 = 5, P_DONE = 6 };
// equality residual (relative to max(1,|q|inf)) below which an un-refined active-set solve is accepted without the refinement solve (the certificate
// itself asks for 1e-6; a solve that misses this tighter bound is refined as before)
#ifndef STALL_WINDOW
#define STALL_WINDOW 6
#endif
#ifndef STALL_MIN_IT
#define STALL_MIN_IT 24
#endif
#ifndef STALL_FACTOR
#define STALL_FACTOR 0.5
#endif
#ifndef AS_DELTA
#define AS_DELTA 1e-13      // diagonal regularisation of the active-set rounds' factorisations
#endif
#ifndef RES_ONLY_TOL
#define RES_ONLY_TOL 1e-9
#endif
struct QpState {   // per instance, 40 doubles
    double phase, it, status, mu, smu, qscale, mtot, pol_round, pol_fail, warm, kst, kbox, ksign, pst, pbox, psign, ticks, fticks, tight, pad;
    double snap_call, snap_mu, snap_used, pad2;   // call that took the iterate copy (0 = none), its mu, 1 = this solve restarted from it
    double mode, cold_as, nviol, path;            // mode 1: P_INIT starts the interior point (0: an active-set attempt from the empty set);
                                                  // cold_as 1: that attempt was made; nviol: violated bounds seen by the previous round;
                                                  // path: how the solve ended up where it is (qp_stats[7])
    unsigned long long seth[4];                   // hashes of the last active sets of the current attempt (a repeat = the iteration cycles)
    double kmin, fact_call, act1_ok, uf_valid;    // first stage whose block D_k the next P_POL0 factorisation must recompute (see ne_forward k0);
                                                  // fact_call: call whose last certified solve left its factorisation (for the set in ACT) in the
                                                  // scratch; act1_ok: ACT1 holds a certified set; uf_valid: the last forward sweep solved the current attempt's
                                                  // un-refined system, so UF's leading u_k carry over to the next round (ne_forward ks)
    double res_only, tbox, stall_ref, stall_it;            // res_only 1: the look at the un-refined solve already found stationarity, box and multiplier signs within the
                                                  // certificate's tolerance (tbox: its box violation); the tick that follows only has to measure the equality
                                                  // residual -- if that is small too the un-refined solve IS the certified answer and the refinement is skipped;
                                                  // stall_ref / stall_it: max(residual, mu) of the interior point at its last checkpoint (every 6 iterations)
};
static_assert(sizeof(QpState) == 40 * sizeof(double), "QpState size");

#ifndef LAUNDER_B
#define LAUNDER_B(b) asm volatile("" : "+s"(b))
#endif
template <int NX, int NU>
__device__ __forceinline__ NeG<NX, NU> make_neg(const QpArgs &a, int b) {
    using L = Lay<NX, NU>;
    const int N = a.N, n = L::n(N), mb = L::mb(N);
    double *ws = a.ws + (size_t)b * qp_ws_doubles(n, N, NX);
    NeG<NX, NU> g;
    g.A = a.A + (size_t)b * N * NX * NX; g.Bm = a.Bm + (size_t)b * N * NX * NU;
    g.ub = a.ubg + (size_t)b * mb; g.lb = a.lbg + (size_t)b * mb;
    g.Linv = a.Linv + (size_t)b * N * NX * NX; g.PI = ws + 8 * (size_t)n; g.V = ws + 9 * (size_t)n; g.G = ws + 10 * (size_t)n;
    g.W = ws + 12 * (size_t)n; g.UF = ws + 19 * (size_t)n + 4 * (size_t)N * NX; g.N = N;
    return g;
}

// what the forward sweep of a tick does, from the instance's phase: factorise? right-hand side with the dynamics offsets? regularisation; first
// stage to re-factorise.  The active-set rounds regularise with 1e-10 (also the P_INIT solve that serves as their round 0, so that the next
// round can keep its leading blocks).
struct FwdPlan { bool factor; double eflag, delta; int k0, ks; };
__device__ __forceinline__ FwdPlan fwd_plan(const QpState *st, int phase, int N) {
    FwdPlan p;
    p.factor = (phase == P_INIT || phase == P_PRED || phase == P_POL0);
    p.eflag = (phase == P_INIT || phase >= P_POL0) ? 1.0 : 0.0;
    p.delta = (phase == P_POL0 || (phase == P_INIT && st->mode == 0.0)) ? AS_DELTA : 0.0;
    p.k0 = (phase == P_POL0) ? (int)st->kmin : 0;
    p.ks = (phase == P_POL0 && st->uf_valid != 0.0 && p.k0 < N) ? p.k0 : 0;   // (kmin = N: first tick of a QP that inherits its factorisation -- its right-hand side is new)
    return p;
}

// first = 1: set up the instance (x0-pin check, starting rhs); else consume the solve of the current phase.
template <int NX, int NU>
__device__ __forceinline__ void phase_update(const QpArgs &a, int first, int b, int lane, double *sm = nullptr) {
    using L = Lay<NX, NU>;
    constexpr int NZ = L::NZ, SR = L::SR;
    const int N = a.N, n = L::n(N), mb = L::mb(N);
    const double *ub = a.ubg + (size_t)b * mb, *qg = a.q + (size_t)b * n;
    double *ws = a.ws + (size_t)b * qp_ws_doubles(n, N, NX);
    double *Z = ws, *SU = Z + n, *SL = SU + n, *LU = SL + n, *LL = LU + n, *GC = LL + n, *CU = GC + n, *CL = CU + n;
    double *PI = CL + n, *V = PI + n, *G = V + n, *ACT = G + n, *W = ACT + n, *NUA = W + N * NX, *NUP = NUA + N * NX;
    double *SZ = NUP + N * NX, *SSU = SZ + n, *SSL = SSU + n, *SLU = SSL + n, *SLL = SLU + n, *SGC = SLL + n, *SNUA = SGC + n;   // copy of an interior-point iterate
    double *ACT1 = SNUA + N * NX;   // active set the first QP of the previous fast-SLS call ended on (warm start of the next call's first QP)
    double *ACT2 = ws + 19 * (size_t)n + 5 * (size_t)N * NX;   // ... and the one its last QP ended on (behind UF; warm start of the next call's last QP, QpArgs::as_warm_last)
    Costs cst = a.cst;
    cst.prox = a.prox ? a.prox[(size_t)b * a.prox_stride] : 0.0;
    const double call_id = a.call_ids ? a.call_ids[b] : a.call_id;
    QpState *stp = (QpState *)a.state + b;
    // Restart of the interior point from the copy taken by an earlier QP of the same fast-SLS call (same A, B, q; the bounds moved): primal and
    // multipliers are kept, a slack that the new bound would make smaller than min(s, max(sqrt(mu), 1e-3)) is pushed back to that floor.  On the
    // closed-loop rocket QPs the tightened QP then needs 3-5 iterations instead of 6-7 (scripts/proto/ipm_warm_stats.py).
    auto restore_iterate = [&](double mu_snap) {
        const double smin = fmax(sqrt(fmax(mu_snap, 0.0)), 1e-3);
#pragma unroll 4
        for (int e = lane; e < n; e += 64) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            const double z = (e < NX) ? a.x0val[(size_t)b * NX + e] : SZ[e], su = SSU[e], sl = SSL[e];
            Z[e] = z; GC[e] = SGC[e];
            SU[e] = el.fu ? fmax(el.hi - z, fmin(su, smin)) : 1.0; LU[e] = el.fu ? SLU[e] : 0.0;
            SL[e] = el.fl ? fmax(z - el.lo, fmin(sl, smin)) : 1.0; LL[e] = el.fl ? SLL[e] : 0.0;
        }
        for (int o = lane; o < N * NX; o += 64) NUA[o] = SNUA[o];
    };

    if (first) {
        int status = ST_INIT;
        double viol = 0.0;   // x0 pin vs its own box (the reference applies both the pin rows and the stage-0 inequality rows)
        if (lane < NX) {
            const double xv = a.x0val[(size_t)b * NX + lane];
            viol = fmax(xv - ub[NX + lane], -ub[NX + NZ + lane] - xv);
        }
        if (wla::wave_max(viol) > 1e-9) status = 2;
        // warm start: the previous solve of this instance ended with a certified active set -> polish from it first
        // the first QP of a call starts from the set the first QP of an earlier call ended on (the un-tightened QPs of consecutive MPC steps
        // resemble each other more than a tightened and an un-tightened one: 2.8 against 3.9 rounds, scripts/proto/as_warm_sources.py) -- also
        // when the solve in between did not end on a certificate (e.g. a measured state outside its own box, status 2): ACT1 is only ever
        // written with a certified set
        // act1_ok packs: bit 0 ACT1 holds a certified set, bit 1 ACT2 does; bits 2..9 / 10..17 the horizon shifts since ACT1 / ACT2 were written.  A set is
        // moved by as many stages as the horizon has moved since it was certified -- not by one: a step that is flagged at x0 (or whose solve fails)
        // writes no set, and the next step's QP would otherwise start from a set that is one stage off (the rocket loop from the script's x0 flags
        // 5-20 % of the steps; their successors were most of the warm attempts that failed)
        int okbits = (int)stp->act1_ok;
        int age1 = (okbits >> 2) & 255, age2 = (okbits >> 10) & 255;
        if (a.stat_slot == 0 && a.snap_use == 0 && a.warm_shift && (!a.shift_stepno || a.shift_stepno[b] > 0)) { age1 = min(age1 + 1, 255); age2 = min(age2 + 1, 255); }      // the call's first QP sees the shift once
        okbits = (okbits & 3) | (age1 << 2) | (age2 << 10);
        const bool from_act1 = a.warm && status == ST_INIT && a.stat_slot == 0 && a.snap_use == 0 && (okbits & 1);
        bool warm = a.warm && status == ST_INIT && (from_act1 || (((int)stp->status == 0) && ((int)stp->phase == P_DONE)));
        // The last QP of a call (tightened bounds).  Its natural starting set is the first QP's (same A, B, q, factorisation at hand), and up to
        // ~28 active bounds that is the cheapest start.  Beyond, the tightening moves so many touch points that the rounds cost more than an
        // interior point -- but the set the LAST QP of the previous MPC step ended on, moved one stage with the horizon, is nearly right: on the
        // closed loop from the script's x0 (34-39 active bounds) 4.2 rounds and no failure in 88 QPs, against 11.1 rounds and 8 failures from
        // the first QP's set (scripts/proto/as_warm_sources.py on QPs dumped from the GPU loop).
        const bool have_act2 = a.warm && status == ST_INIT && a.stat_slot == 1 && a.as_warm_last > 0 && (okbits & 2);
        bool from_act2 = have_act2 && (a.as_warm_last >= 2 || !warm);
        bool big_set = false;
        if (warm && !from_act1 && !from_act2 && a.snap_use != 0 && a.as_warm_max_set > 0) {
            double cnt = 0.0;
            for (int e = lane; e < n; e += 64) cnt += (ACT[e] != 0.0) ? 1.0 : 0.0;
            big_set = wla::wave_sum(cnt) > (double)a.as_warm_max_set;
            if (big_set) { if (have_act2) { from_act2 = true; big_set = false; } else warm = false; }
        }
        if (from_act2) warm = true;
        const double *prev = a.primal + (size_t)b * n;
        // a later QP of the same call whose previous solve was certified: same A, B, weights, and the scratch still holds the factorisation of
        // exactly the set it starts from -- its first tick needs no factorisation at all
        const bool keep_fact = warm && !from_act1 && !from_act2 && a.snap_use != 0 && stp->fact_call == call_id && call_id != 0.0;
        double set_changed = 0.0;
        double qscale = 0.0, mtot = 0.0;
#pragma unroll 4
        for (int e = lane; e < n; e += 64) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            qscale = fmax(qscale, fabs(el.q));
            mtot += (el.fu ? 1.0 : 0.0) + (el.fl ? 1.0 : 0.0);
            if (warm) {
                const double ac_old = ACT[e];
                const double *src = from_act1 ? ACT1 : ACT2;
                int es = e + (from_act1 ? age1 : age2) * NZ;      // the same component, as many stages later as the horizon moved; past the end: the last stage that has it
                while (es >= n) es -= NZ;
                double ac = (from_act1 || from_act2) ? src[es] : ac_old;
                if ((ac > 0.0 && !el.fu) || (ac < 0.0 && !el.fl) || !el.fr) ac = 0.0;
                if (ac != ac_old) set_changed = 1.0;
                const double zp = (e < NX) ? a.x0val[(size_t)b * NX + e] : prev[e];
                const double z0 = ac > 0.0 ? el.hi : (ac < 0.0 ? el.lo : zp);
                const double pi = (el.fr && ac == 0.0) ? wla::fast_rcp(el.pd) : 0.0;
                ACT[e] = ac; PI[e] = pi; V[e] = (pi != 0.0) ? -pi * el.q : z0;
            } else {
                const double z0 = (e < NX) ? a.x0val[(size_t)b * NX + e] : 0.0;   // rhs of P_INIT: v = z0 - Pi (P z0 + q), z0 = [x0;0]
                const double pi = el.fr ? wla::fast_rcp(el.pd) : 0.0;
                PI[e] = pi; V[e] = (pi != 0.0) ? -pi * el.q : z0; Z[e] = z0;
            }
        }
        qscale = fmax(1.0, wla::wave_max(qscale));
        mtot = fmax(1.0, wla::wave_sum(mtot));
        const bool skip_fact = keep_fact && wla::wave_max(set_changed) == 0.0;
        if (lane == 0) {
            QpState s0;
            s0.phase = (status == ST_INIT) ? (warm ? P_POL0 : P_INIT) : P_DONE; s0.it = 0; s0.status = status; s0.mu = 0; s0.smu = 0; s0.qscale = qscale;
            s0.mtot = mtot; s0.pol_round = 0; s0.pol_fail = 0; s0.warm = warm ? 1.0 : 0.0; s0.kst = 0; s0.kbox = 0; s0.ksign = 0; s0.pst = -1; s0.pbox = -1; s0.psign = -1; s0.ticks = 0; s0.fticks = 0; s0.tight = 0; s0.pad = 0;
            s0.snap_call = stp->snap_call; s0.snap_mu = stp->snap_mu; s0.snap_used = 0; s0.pad2 = 0;
            s0.mode = (a.as_first && !big_set) ? 0.0 : 1.0; s0.cold_as = big_set ? 1.0 : 0.0; s0.nviol = 0; s0.path = warm ? (from_act2 ? 20.0 : 10.0) : (big_set ? 1.0 : 0.0);
            s0.seth[0] = s0.seth[1] = s0.seth[2] = s0.seth[3] = 0ULL; s0.kmin = skip_fact ? (double)N : 0.0; s0.fact_call = 0; s0.act1_ok = (double)okbits; s0.uf_valid = 0; s0.res_only = 0; s0.tbox = 0; s0.stall_ref = 0; s0.stall_it = 0;
            *stp = s0;
            a.status[b] = status; a.iters[b] = 0;
#ifdef QP_DIAG_SPAN
            if (a.diag) for (int i = 0; i < 32; i++) a.diag[((size_t)b * 2 + a.stat_slot) * 32 + i] = -1;
#endif
            if (status != ST_INIT && a.qpstat) {   // flagged without a solve: the statistics of this slot must not show the previous call's
                int *qs = a.qpstat + ((size_t)b * 2 + a.stat_slot) * 8;
                for (int i = 0; i < 8; i++) qs[i] = 0;
                qs[6] = status;
            }
        }
        return;
    }

    QpState s = *stp;
    int phase = (int)s.phase;
    if (phase == P_DONE) return;
    int status = (int)s.status, it = (int)s.it;
    const double qscale = s.qscale, mtot = s.mtot, ptol = 1e-9 * qscale;
    double tol = (s.tight != 0.0 ? fmin(a.eps, 1e-9) : a.eps) * qscale;
    bool polished = false, start_iter = false, fall_back = false;

    bool as_from_init = false;
    if (phase == P_INIT && s.mode == 0.0) {
        // active-set attempt from the empty set: the solve just made is its round 0 (all bounds inactive); hand it to the polish logic below
#pragma unroll 4
        for (int e = lane; e < n; e += 64) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            const double gc = G[e];
            CU[e] = el.fr ? -(el.q + gc) * wla::fast_rcp(el.pd) : Z[e];
            CL[e] = gc; ACT[e] = 0.0;
        }
        for (int o = lane; o < N * NX; o += 64) NUP[o] = W[o];
        s.cold_as = 1.0; s.warm = 2.0; s.pol_round = 0.0; s.pol_fail = 0.0; s.nviol = 0.0;
        as_from_init = true;
        phase = P_POL0;
        wla::wsync_mem();
    }
    if (phase == P_INIT) {
        for (int o = lane; o < N * NX; o += 64) NUA[o] = W[o];
#pragma unroll 4
        for (int e = lane; e < n; e += 64) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            const double gc = G[e];
            const double z = el.fr ? -(el.q + gc) * wla::fast_rcp(el.pd) : Z[e];
            GC[e] = gc; Z[e] = z;
            // starting point: slacks floored at init_s, multipliers init_lam (<= 0: scaled with the linear cost, max(1, 0.1 |q|inf))
            const double lam0 = a.init_lam > 0.0 ? a.init_lam : fmax(1.0, 0.1 * qscale);
            SU[e] = el.fu ? fmax(el.hi - z, a.init_s) : 1.0; LU[e] = el.fu ? lam0 : 0.0;
            SL[e] = el.fl ? fmax(z - el.lo, a.init_s) : 1.0; LL[e] = el.fl ? lam0 : 0.0;
        }
        start_iter = true;
    } else if (phase == P_PRED) {
        // affine step: step length, mu_aff = (S0 + a S1 + a^2 S2)/m, second-order terms ds*dlambda
        double amin = 1.0, S0 = 0.0, S1 = 0.0, S2 = 0.0;
#pragma unroll 4
        for (int e = lane; e < n; e += 64) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            const double dz = V[e] - PI[e] * G[e];
            const double z = Z[e], su = SU[e], sl = SL[e], lu = LU[e], ll = LL[e];
            double cu = 0.0, cl = 0.0;
            if (el.fu) {
                const double ru = z + su - el.hi, dsu = -ru - dz, dlu = -lu + (lu * ru + lu * dz) * wla::fast_rcp(su);
                if (dsu < 0) amin = fmin(amin, -su * wla::fast_rcp(dsu));
                if (dlu < 0) amin = fmin(amin, -lu * wla::fast_rcp(dlu));
                S0 += su * lu; S1 += su * dlu + lu * dsu; cu = dsu * dlu; S2 += cu;
            }
            if (el.fl) {
                const double rl = el.lo - z + sl, dsl = -rl + dz, dll = -ll + (ll * rl - ll * dz) * wla::fast_rcp(sl);
                if (dsl < 0) amin = fmin(amin, -sl * wla::fast_rcp(dsl));
                if (dll < 0) amin = fmin(amin, -ll * wla::fast_rcp(dll));
                S0 += sl * ll; S1 += sl * dll + ll * dsl; cl = dsl * dll; S2 += cl;
            }
            CU[e] = cu; CL[e] = cl;
        }
        const double aaff = wla::wave_min(amin);
        S0 = wla::wave_sum(S0); S1 = wla::wave_sum(S1); S2 = wla::wave_sum(S2);
        const double muaff = (S0 + aaff * (S1 + aaff * S2)) / mtot;
        double sig = muaff / s.mu; sig = sig * sig * sig;
        const double smu = sig * s.mu;
        s.smu = smu;
#pragma unroll 4
        for (int e = lane; e < n; e += 64) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            const double z = Z[e], su = SU[e], sl = SL[e], lu = LU[e], ll = LL[e];
            double rr = el.fr ? el.pd * z + el.q + GC[e] + lu - ll : 0.0;
            if (el.fu) rr += (-(su * lu + CU[e] - smu) + lu * (z + su - el.hi)) * wla::fast_rcp(su);
            if (el.fl) rr -= (-(sl * ll + CL[e] - smu) + ll * (el.lo - z + sl)) * wla::fast_rcp(sl);
            V[e] = -PI[e] * rr;
        }
        phase = P_CORR;
    } else if (phase == P_CORR) {
        const double smu = s.smu;
        double amin = 1e300;
#pragma unroll 4
        for (int e = lane; e < n; e += 64) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            const double dz = V[e] - PI[e] * G[e];
            const double z = Z[e], su = SU[e], sl = SL[e], lu = LU[e], ll = LL[e];
            if (el.fu) {
                const double ru = z + su - el.hi, dsu = -ru - dz;
                const double dlu = (-(su * lu + CU[e] - smu) + lu * ru + lu * dz) * wla::fast_rcp(su);
                if (dsu < 0) amin = fmin(amin, -su * wla::fast_rcp(dsu));
                if (dlu < 0) amin = fmin(amin, -lu * wla::fast_rcp(dlu));
            }
            if (el.fl) {
                const double rl = el.lo - z + sl, dsl = -rl + dz;
                const double dll = (-(sl * ll + CL[e] - smu) + ll * rl - ll * dz) * wla::fast_rcp(sl);
                if (dsl < 0) amin = fmin(amin, -sl * wla::fast_rcp(dsl));
                if (dll < 0) amin = fmin(amin, -ll * wla::fast_rcp(dll));
            }
        }
        const double alpha = fmin(1.0, 0.99 * wla::wave_min(amin));
#pragma unroll 4
        for (int e = lane; e < n; e += 64) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            const double gg = G[e];
            const double dz = V[e] - PI[e] * gg;
            const double z = Z[e], su = SU[e], sl = SL[e], lu = LU[e], ll = LL[e];
            if (el.fu) {
                const double ru = z + su - el.hi, dsu = -ru - dz;
                const double dlu = (-(su * lu + CU[e] - smu) + lu * ru + lu * dz) * wla::fast_rcp(su);
                SU[e] = su + alpha * dsu; LU[e] = lu + alpha * dlu;
            }
            if (el.fl) {
                const double rl = el.lo - z + sl, dsl = -rl + dz;
                const double dll = (-(sl * ll + CL[e] - smu) + ll * rl - ll * dz) * wla::fast_rcp(sl);
                SL[e] = sl + alpha * dsl; LL[e] = ll + alpha * dll;
            }
            Z[e] = z + alpha * dz; GC[e] += alpha * gg;
        }
        for (int o = lane; o < N * NX; o += 64) NUA[o] += alpha * W[o];
        it++;
        start_iter = true;
    } else if (phase == P_POL2 && s.res_only != 0.0 && stp->pbox < RES_ONLY_TOL * qscale) {
        // the un-refined solve met every condition of the certificate (element-wise ones checked by the look, equality residual just measured by
        // the forward sweep): it is the answer.  k_qp_solve made the same test and did not run the backward sweep; CU, CL, NUP hold the solve.
        polished = true; status = 0; s.kst = s.pst; s.kbox = s.tbox; s.ksign = s.psign; s.pbox = stp->pbox; s.res_only = 0.0;
        phase = P_DONE;
    } else {
        s.res_only = 0.0;
        // polish phases: zn = v - Pi g  (v was z0 - Pi r or zn - Pi r1); CU := zn, CL := accumulated E' nu
        const bool firstp = (phase == P_POL0);
        const double max_rounds = (s.warm == 1.0) ? (double)a.warm_rounds : (s.warm == 2.0 ? (double)a.as_rounds : 8.0);
        // the look at an un-refined solve (the common tick of an active-set attempt) runs fused below when the wave's LDS is at hand
        const bool look = (phase == P_POL0 && s.pol_fail == 0.0 && s.pol_round < max_rounds);
        const bool fused_look = look && sm != nullptr;
        if (!as_from_init && !fused_look) {
#pragma unroll 4
            for (int e = lane; e < n; e += 64) {
                const double gg = G[e];
                CU[e] = V[e] - PI[e] * gg;
                CL[e] = (firstp ? 0.0 : CL[e]) + gg;
            }
            for (int o = lane; o < N * NX; o += 64) NUP[o] = (firstp ? 0.0 : NUP[o]) + W[o];
            wla::wsync_mem();
        }
        // Correction of the active set from the solve in CU (primal) / CL (E'nu): multipliers of the wrong sign leave, violated bounds enter --
        // all of them for the inputs (control constraints are active on arcs), but for a state component only the stages where its violation
        // has a local maximum along the horizon: state constraints are active at isolated touch points, and fixing a whole violated arc at once
        // over-constrains the next solve and sets the iteration oscillating (measured on closed-loop rocket QPs: every warm attempt failed with
        // the add-all rule, 3-5 rounds suffice with this one, scripts/proto/as_localmax.py).  Pass 1 writes the new set into G (free until the
        // next backward sweep) because the rule reads neighbouring elements; pass 2 applies it.  Returns changes made; nv = violated bounds seen.
        auto viol_at = [&](int e, double tolv) -> double {
            if (e < NX || e >= n || ACT[e] != 0.0) return 0.0;
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            const double zn = CU[e];
            double v = 0.0;
            if (el.fu) v = fmax(v, zn - el.hi);
            if (el.fl) v = fmax(v, el.lo - zn);
            return v > tolv ? v : 0.0;
        };
        auto plan_set = [&](double tolv, double &nv, unsigned long long &hash) -> double {
            // inputs first: a state that leaves its box because the input driving it does (actuator lags: servo angle / servo command) must not
            // be pinned while that input is still free -- the next solve would swing the inputs by orders of magnitude.  Bounds on states enter
            // only in rounds that find no violated input bound.
            int in_viol = 0;
            for (int o = lane; o < N * NU; o += 64) {
                const int e = (o / NU) * NZ + NX + (o % NU);
                in_viol |= (viol_at(e, tolv) > 0.0) ? 1 : 0;
            }
            const bool any_in = wla::wave_or(in_viol) != 0;
            double changed = 0.0, viols = 0.0, kfirst = 1e9;
            unsigned long long hv = 0ULL;
#pragma unroll 2
            for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double zn = CU[e], gr = el.pd * zn + el.q + CL[e], ac = ACT[e];
                double nac = ac;
                if (ac > 0.0 && gr > tolv) nac = 0.0;
                if (ac < 0.0 && -gr > tolv) nac = 0.0;
                if (ac == 0.0 && el.fr) {
                    const double vu = el.fu ? zn - el.hi : 0.0, vl = el.fl ? el.lo - zn : 0.0, v = fmax(vu, vl);
                    if (v > tolv) {
                        viols += 1.0;
                        bool take = true;
                        if ((e % NZ) < NX) take = !any_in && (v >= viol_at(e - NZ, tolv)) && (v >= viol_at(e + NZ, tolv));
                        if (take) nac = vu > vl ? 1.0 : -1.0;
                    }
                }
                if (nac != ac) { changed += 1.0; kfirst = fmin(kfirst, (double)max(0, e / NZ - ((e % NZ) < NX ? 1 : 0))); }   // x_k enters D_{k-1} and D_k, u_k enters D_k
                if (nac != 0.0) {   // splitmix64 of (element, side)
                    unsigned long long x = (unsigned long long)(2 * e + (nac > 0.0 ? 1 : 0)) + 0x9E3779B97F4A7C15ULL;
                    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; hv ^= x ^ (x >> 31);
                }
                G[e] = nac;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned lo32 = __shfl_xor((unsigned)(hv & 0xFFFFFFFFULL), o), hi32 = __shfl_xor((unsigned)(hv >> 32), o);
                hv ^= ((unsigned long long)hi32 << 32) | lo32;
            }
            hash = hv | 1ULL;      // never 0 (0 = empty slot)
            nv = wla::wave_sum(viols);
            s.kmin = fmin(wla::wave_min(kfirst), (double)N);
            return wla::wave_sum(changed);
        };
        // a set this attempt has already solved with: the iteration cycles
        auto seen_before = [&](unsigned long long hsh) -> bool {
            const bool rep = (hsh == s.seth[0]) || (hsh == s.seth[1]) || (hsh == s.seth[2]) || (hsh == s.seth[3]);
            s.seth[3] = s.seth[2]; s.seth[2] = s.seth[1]; s.seth[1] = s.seth[0]; s.seth[0] = hsh;
            return rep;
        };
        auto apply_set = [&]() {
            wla::wsync_mem();
#pragma unroll 4
            for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double zn = CU[e], ac = G[e];
                const double z0 = ac > 0.0 ? el.hi : (ac < 0.0 ? el.lo : zn);
                const double pi = (el.fr && ac == 0.0) ? wla::fast_rcp(el.pd) : 0.0;
                ACT[e] = ac; PI[e] = pi; V[e] = (pi != 0.0) ? -pi * el.q : z0;
            }
        };
        bool again = false, give_up = false;
        // The same look in one trip to memory: every global array is read once (pass A), the primal zn and E'nu go back out, each element's
        // violation and its set entry after the releases go to LDS (free between the sweeps), the neighbour tests of the local-maximum rule
        // read them there (pass B), and the new set is applied from LDS (pass C).  The unfused version below makes five dependent trips
        // (update zn, input pre-pass, plan, apply, ...): 45 us per tick with 3 waves per SIMD contending for memory, against 15 us alone.
        double tvst = 0.0, tvbox = 0.0, tvsign = 0.0;      // the certificate's quantities of the un-refined solve (fused look only)
        auto fused_plan = [&](double tolv, double &nv, unsigned long long &hash) -> double {
            double *sV = sm, *sN = sm + n;
            int in_viol = 0;
            double viols = 0.0;
#pragma unroll 2
            for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                double zn, cl;
                if (as_from_init) { zn = CU[e]; cl = CL[e]; }
                else {
                    const double gg = G[e];
                    zn = V[e] - PI[e] * gg; cl = (firstp ? 0.0 : CL[e]) + gg;
                    CU[e] = zn; CL[e] = cl;
                }
                const double gr = el.pd * zn + el.q + cl, ac = ACT[e];
                if (el.fr && ac == 0.0) tvst = fmax(tvst, fabs(gr));
                if (el.fu) tvbox = fmax(tvbox, zn - el.hi);
                if (el.fl) tvbox = fmax(tvbox, el.lo - zn);
                if (ac > 0.0) tvsign = fmax(tvsign, gr);
                if (ac < 0.0) tvsign = fmax(tvsign, -gr);
                double nac = ac;
                if (ac > 0.0 && gr > tolv) nac = 0.5;          // 0.5: released in this round (reads as "not in the set" below, counts as a change)
                if (ac < 0.0 && -gr > tolv) nac = 0.5;
                double v = 0.0;
                if (ac == 0.0 && el.fr) {
                    const double vu = el.fu ? zn - el.hi : 0.0, vl = el.fl ? el.lo - zn : 0.0, vm = fmax(vu, vl);
                    if (vm > tolv) { v = vu > vl ? vm : -vm; viols += 1.0; if ((e % NZ) >= NX) in_viol = 1; }
                }
                sV[e] = v; sN[e] = nac;
            }
            if (!as_from_init) for (int o = lane; o < N * NX; o += 64) NUP[o] = (firstp ? 0.0 : NUP[o]) + W[o];
            wla::wsync();
            const bool any_in = wla::wave_or(in_viol) != 0;
            double changed = 0.0, kfirst = 1e9;
#ifdef QP_DIAG_SPAN
            double klast = -1.0;
#endif
            unsigned long long hv = 0ULL;
#pragma unroll 2
            for (int e = lane; e < n; e += 64) {
                const double v = sV[e];
                double nac = sN[e];
                bool chg = (nac == 0.5);
                if (chg) nac = 0.0;
                if (v != 0.0) {
                    bool take = true;
                    if ((e % NZ) < NX) {
                        const double va = fabs(v), vp = (e - NZ >= NX) ? fabs(sV[e - NZ]) : 0.0, vn = (e + NZ < n) ? fabs(sV[e + NZ]) : 0.0;
                        take = !any_in && va >= vp && va >= vn;
                    }
                    if (take) { nac = v > 0.0 ? 1.0 : -1.0; chg = true; }
                }
                if (chg) { changed += 1.0; kfirst = fmin(kfirst, (double)max(0, e / NZ - ((e % NZ) < NX ? 1 : 0))); }
#ifdef QP_DIAG_SPAN
                if (chg) klast = fmax(klast, (double)min(N - 1, e / NZ));
#endif
                if (nac != 0.0) {
                    unsigned long long x = (unsigned long long)(2 * e + (nac > 0.0 ? 1 : 0)) + 0x9E3779B97F4A7C15ULL;
                    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; hv ^= x ^ (x >> 31);
                }
                // (own element only: no other lane reads sN)
                sN[e] = nac;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned lo32 = __shfl_xor((unsigned)(hv & 0xFFFFFFFFULL), o), hi32 = __shfl_xor((unsigned)(hv >> 32), o);
                hv ^= ((unsigned long long)hi32 << 32) | lo32;
            }
            hash = hv | 1ULL;
            nv = wla::wave_sum(viols);
            s.kmin = fmin(wla::wave_min(kfirst), (double)N);
#ifdef QP_DIAG_SPAN
            {
                const double kl = wla::wave_max(klast);
                const int r = (int)s.pol_round;
                if (lane == 0 && a.diag && r < 16 && kl >= 0.0) { int *dg = a.diag + (((size_t)b * 2 + a.stat_slot) * 16 + r) * 2; dg[0] = (int)s.kmin; dg[1] = (int)kl; }
            }
#endif
            return wla::wave_sum(changed);
        };
        auto fused_apply = [&]() {
            const double *sN = sm + n;
#pragma unroll 4
            for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double ac = sN[e];
                const double z0 = ac > 0.0 ? el.hi : (ac < 0.0 ? el.lo : 0.0);
                const double pi = (el.fr && ac == 0.0) ? wla::fast_rcp(el.pd) : 0.0;
                ACT[e] = ac; PI[e] = pi; V[e] = (pi != 0.0) ? -pi * el.q : ((e < NX) ? a.x0val[(size_t)b * NX + e] : z0);
            }
        };
        if (look) {
            // cheap look at the un-refined solve: if the active set is visibly wrong (coarse tolerance) correct it now and
            // factorise again, without spending the refinement solve on a set that is about to change
            double nv = 0.0;
            unsigned long long hsh = 0ULL;
            const double changed = fused_look ? fused_plan(a.early_ctol * qscale, nv, hsh) : plan_set(a.early_ctol * qscale, nv, hsh);
            again = changed > 0.0;
            if (again && s.warm > 0.0 && seen_before(hsh)) { again = false; give_up = true; }
            // an attempt (warm or from the empty set) whose solve blows up -- a set that pins both ends of a dynamics row leaves hundreds of
            // violated bounds -- is left to the interior point at once
            // (the relative rule only from 24 violated bounds on: a round that releases a dozen multipliers of the wrong sign may well go from 2 to 13
            // violated bounds and still converge in eight rounds -- closed-loop step 6 from the script's x0 lost 92 % of its first QPs to it)
            if (s.warm > 0.0 && (nv > (double)a.as_max_viol || (s.pol_round > 0.0 && nv > 2.0 * s.nviol + 8.0 && nv > 24.0))) { again = false; give_up = true; }
            s.nviol = nv;
            if (again) { s.pol_round += 1.0; if (fused_look) fused_apply(); else apply_set(); s.uf_valid = 1.0; }      // the sweep just consumed solved the un-refined system of this attempt
            if (fused_look) wla::wsync_mem();
        } else if (phase == P_POL0 && s.warm > 0.0) give_up = true;      // out of rounds (or a pivot failed): do not refine a set known to be wrong
        if (again) {
            phase = P_POL0;
        } else if (give_up) {
            fall_back = true;
        } else if (phase != P_POL2) {
    #pragma unroll 4
        for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double pi = PI[e], zn = CU[e];
                const double r1 = (pi != 0.0) ? el.pd * zn + el.q + CL[e] : 0.0;
                V[e] = zn - pi * r1;
            }
            s.res_only = 0.0;
            if (fused_look && a.n_refine <= 1 && s.pol_fail == 0.0) {
                // fp64: the un-refined solve is normally accurate to rounding.  If it already meets the certificate's element-wise conditions, the next
                // tick's forward sweep measures its equality residual and, when that is tiny too, nothing is left to refine (k_qp_solve skips the
                // backward sweep; the accept test is at the top of the polish branch)
                const double wst = wla::wave_max(tvst), wbox = wla::wave_max(tvbox), wsign = wla::wave_max(tvsign);
                if (wst < ptol && wbox < ptol && wsign < ptol) { s.res_only = 1.0; s.pst = wst; s.tbox = wbox; s.psign = wsign; }
            }
            // n_refine refinement solves per polish: P_POL1 repeats until the last one, which runs as P_POL2 (s.pad counts them)
            if (phase == P_POL0) { s.pad = 0.0; phase = (a.n_refine <= 1) ? P_POL2 : P_POL1; }
            else { s.pad += 1.0; phase = (s.pad + 1.0 >= (double)a.n_refine) ? P_POL2 : P_POL1; }
        } else {
            double vst = 0.0, vbox = 0.0, vsign = 0.0;
    #pragma unroll 4
        for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double zn = CU[e], gr = el.pd * zn + el.q + CL[e], ac = ACT[e];
                if (el.fr && ac == 0.0) vst = fmax(vst, fabs(gr));
                if (el.fu) vbox = fmax(vbox, zn - el.hi);
                if (el.fl) vbox = fmax(vbox, el.lo - zn);
                if (ac > 0.0) vsign = fmax(vsign, gr);     // lambda_u = -gr must be >= 0
                if (ac < 0.0) vsign = fmax(vsign, -gr);    // lambda_l = +gr must be >= 0
            }
            vst = wla::wave_max(vst); vbox = wla::wave_max(vbox); vsign = wla::wave_max(vsign);
            s.pst = vst; s.psign = vsign;
            const bool pf = s.pol_fail != 0.0;
            const double eqres = stp->pbox;            // max |E zn - e| seen by the last refinement's forward sweep (before its correction)
            s.pbox = eqres;
            if (!pf && vst < ptol && vbox < ptol && vsign < ptol && eqres < 1e-6 * qscale) {
                polished = true; status = 0; s.kst = vst; s.kbox = vbox; s.ksign = vsign;
                phase = P_DONE;
            } else if (!pf && vst < ptol && s.pol_round < max_rounds) {
                // primal-dual active-set correction on the refined solve (tight tolerance), factorise again
                double nv = 0.0;
                unsigned long long hsh = 0ULL;
                plan_set(ptol, nv, hsh);
                if (s.warm > 0.0 && seen_before(hsh)) fall_back = true;
                else { s.pol_round += 1.0; s.nviol = nv; apply_set(); phase = P_POL0; s.uf_valid = 0.0; }   // the last sweep was the refinement's
            } else if (s.warm > 0.0) {
                fall_back = true;      // warm / cold active-set attempt failed
            } else if (s.tight == 0.0 && a.eps > 1e-9 && status == 4) {
                // the interior point stopped at a loose tolerance and its active-set guess did not certify: resume it (its state
                // arrays are untouched by the polish) down to 1e-9, then polish again
                s.tight = 1.0; s.pol_round = 0.0; s.pol_fail = 0.0; tol = fmin(a.eps, 1e-9) * qscale;
                start_iter = true;
            } else phase = P_DONE;
        }
    }

    if (fall_back) {
        // an active-set attempt is abandoned.  A warm one is followed by the attempt from the empty set (when enabled and not yet made), then
        // the interior point takes over: from the iterate an earlier QP of this call left behind when there is one (no P_INIT solve then),
        // from its cold start otherwise
        const bool try_cold_as = a.as_first == 1 && s.cold_as == 0.0 && s.warm == 1.0;   // (as_first 2: a failed warm attempt goes straight to the interior point)
        s.pol_round = 0.0; s.pol_fail = 0.0; s.seth[0] = s.seth[1] = s.seth[2] = s.seth[3] = 0ULL; s.uf_valid = 0.0;
        if (!try_cold_as && a.snap_use && s.snap_call == call_id && call_id != 0.0) {
            s.warm = -1.0; s.path = 10.0 * floor(s.path / 10.0) + 2.0;
            restore_iterate(s.snap_mu);
            s.snap_used = 1.0;
            start_iter = true;
        } else {
            const double w10 = 10.0 * floor(s.path / 10.0);
            if (try_cold_as) { s.mode = 0.0; s.path = w10 + 3.0; s.warm = -1.0; }
            else { s.mode = 1.0; s.path = w10 + 1.0; s.warm = -1.0; }
    #pragma unroll 4
            for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double z0 = (e < NX) ? a.x0val[(size_t)b * NX + e] : 0.0;
                const double pi = el.fr ? wla::fast_rcp(el.pd) : 0.0;
                PI[e] = pi; V[e] = (pi != 0.0) ? -pi * el.q : z0; Z[e] = z0;
            }
            phase = P_INIT;
        }
    }

    if (start_iter) {
        wla::wsync_mem();
        // residuals, complementarity, termination test; predictor rhs written in the same pass
        double rmax = 0.0, musum = 0.0;
#pragma unroll 4
        for (int e = lane; e < n; e += 64) {
            const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
            const double z = Z[e], su = SU[e], sl = SL[e], lu = LU[e], ll = LL[e];
            const double rd = el.fr ? el.pd * z + el.q + GC[e] + lu - ll : 0.0;
            const double ru = el.fu ? z + su - el.hi : 0.0, rl = el.fl ? el.lo - z + sl : 0.0;
            rmax = fmax(rmax, fmax(fabs(rd), fmax(fabs(ru), fabs(rl))));
            musum += (el.fu ? su * lu : 0.0) + (el.fl ? sl * ll : 0.0);
            const double Wu = el.fu ? lu * wla::fast_rcp(su) : 0.0, Wl = el.fl ? ll * wla::fast_rcp(sl) : 0.0;
            const double pi = el.fr ? wla::fast_rcp(el.pd + Wu + Wl) : 0.0;
            const double tu = el.fu ? -lu + Wu * ru : 0.0, tl = el.fl ? -ll + Wl * rl : 0.0;
            PI[e] = pi; V[e] = -pi * (rd + tu - tl);
        }
        const double res = wla::wave_max(rmax);
        const double mu = wla::wave_sum(musum) / mtot;
        s.mu = mu; s.kst = res;
        // Primal infeasibility, certified from the multipliers (what OSQP reports to the reference as "primal infeasible", qp_jit.py:397-400 ->
        // {'success': False}; its test: ||A' dy||inf <= eps ||dy||inf and u'(dy)+ + l'(dy)- < -eps ||dy||inf, eps_prim_inf = 1e-4).  On an infeasible
        // QP the interior point's multipliers y = (nu, lambda_u, lambda_l) diverge along a Farkas ray: E'nu + lambda_u - lambda_l -> 0 relative to
        // |y| while the support value e'nu + hi'lambda_u - lo'lambda_l (x_0 counted as a variable pinned to its value) turns negative.  Measured on
        // the tightened QPs of the closed loop from the script's x0 that the stagnation rule used to end after 24 iterations: |A'y| / |y| falls
        // below 1e-4 at iteration 10-11 with the support value at -2e-3 ... -1e-2 |y|; QPs that solve never get below 0.2 (gpurun_out traces,
        // DESIGN.md section 2.1).  Checked from the 5th iteration on; ends the solve with status 5.
        bool farkas = false;
        if (a.pinf_eps > 0.0 && it >= 4 && res == res) {
            double ymax = 0.0, aty = 0.0, sup = 0.0;
#pragma unroll 4
            for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double lu = el.fu ? LU[e] : 0.0, ll = el.fl ? LL[e] : 0.0, gc = GC[e];
                ymax = fmax(ymax, fmax(lu, ll));
                if (el.fr) { aty = fmax(aty, fabs(gc + lu - ll)); sup += (el.fu ? el.hi * lu : 0.0) - (el.fl ? el.lo * ll : 0.0); }
                else sup -= a.x0val[(size_t)b * NX + e] * gc;          // pinned x_0: multiplier of its pin row = -(E'nu)_0
            }
            const double *lbv = a.lbg + (size_t)b * mb;
            for (int o = lane; o < N * NX; o += 64) {
                const double nu = NUA[o];
                const int r = (o / NX) * SR + (o % NX);
                ymax = fmax(ymax, fabs(nu));
                sup += 0.5 * (ub[r] + lbv[r]) * nu;
            }
            ymax = wla::wave_max(ymax); aty = wla::wave_max(aty); sup = wla::wave_sum(sup);
            farkas = ymax > 0.0 && aty <= a.pinf_eps * ymax && sup < -a.pinf_eps * ymax;
        }
        // first iterate of this solve with mu below snap_mu |q|inf (but not already at the end of the path): keep a copy for the next QP of the call
        if (a.snap_take && s.pad2 == 0.0 && mu <= a.snap_mu * qscale && mu >= 1e-2 * a.snap_mu * qscale && res == res) {
#pragma unroll 4
            for (int e = lane; e < n; e += 64) { SZ[e] = Z[e]; SSU[e] = SU[e]; SSL[e] = SL[e]; SLU[e] = LU[e]; SLL[e] = LL[e]; SGC[e] = GC[e]; }
            for (int o = lane; o < N * NX; o += 64) SNUA[o] = NUA[o];
            s.pad2 = 1.0; s.snap_call = call_id; s.snap_mu = mu;
        }
        // An interior point that no longer makes progress is at an infeasible (or hopelessly ill-posed) QP: once the usual iteration count (7-13) is
        // well behind it (24 iterations), max(residual, mu) must at least halve over six iterations, else the solve is flagged like one that ran out of iterations
        // (status 1) -- 24-30 instead of max_iter = 60 iterations for the stragglers that would otherwise hold the whole launch.
        bool stalled = false;
        {
            const double phi = fmax(res, mu);
            if (s.stall_ref == 0.0) { s.stall_ref = phi; s.stall_it = (double)it; }
            else if ((double)it - s.stall_it >= (double)STALL_WINDOW) {
                stalled = it >= STALL_MIN_IT && phi > STALL_FACTOR * s.stall_ref;
                s.stall_ref = phi; s.stall_it = (double)it;
            }
        }
        if (farkas) { status = 5; phase = P_DONE; }
        else if (!(res == res) || !(mu == mu) || res > 1e30) { status = 3; phase = P_DONE; }
        else if (res < tol && mu < tol) {
            status = 4;
            // polish rhs: active set, z0 with active entries on their bounds, Pi = 0 there
    #pragma unroll 4
        for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                const double z = Z[e], lam = LU[e] - LL[e];
                const bool aU = el.fu && (lam > el.hi - z);
                const bool aL = el.fl && !aU && (-lam > z - el.lo);
                const double z0 = aU ? el.hi : (aL ? el.lo : z);
                const double pi = (el.fr && !aU && !aL) ? wla::fast_rcp(el.pd) : 0.0;
                ACT[e] = aU ? 1.0 : (aL ? -1.0 : 0.0); PI[e] = pi; V[e] = (pi != 0.0) ? -pi * el.q : z0;
            }
            phase = P_POL0; s.kmin = 0.0; s.uf_valid = 0.0;
        } else if (it >= a.max_iter || stalled) {
            // (a solve that had reached the interior-point tolerance and was resumed towards 1e-9 keeps its interior-point-accurate answer)
            status = (s.tight != 0.0 && status == 4) ? 4 : 1; phase = P_DONE;
        }
        else phase = P_PRED;
    }

    if (phase == P_DONE) {
        wla::wsync_mem();
        // ---- write-out (reference layouts: primal qp_jit.py:489-490, duals :493-501) ----
        double *pr = a.primal + (size_t)b * n, *du = a.dual + (size_t)b * mb;
        double csum = 0.0, nact = 0.0;
        const bool ok = (status == 0 || status == 4);   // on failure the previous primal/dual stay (fast_SLS_jit.py:461-464)
        if (ok) {
    #pragma unroll 4
        for (int e = lane; e < n; e += 64) {
                const Elem el = elem_of<NX, NU>(e, n, N, ub, qg, cst);
                double zv, yu, yl, gv;
                if (polished) {
                    zv = CU[e]; gv = CL[e];
                    const double gr = el.pd * zv + el.q + gv, ac = ACT[e];
                    yu = ac > 0.0 ? fmax(-gr, 0.0) : 0.0;
                    yl = ac < 0.0 ? fmax(gr, 0.0) : 0.0;
                } else { zv = Z[e]; gv = GC[e]; yu = el.fu ? LU[e] : 0.0; yl = el.fl ? LL[e] : 0.0; }
                pr[e] = zv;
                nact += (yu > 0.0 ? 1.0 : 0.0) + (yl > 0.0 ? 1.0 : 0.0);
                csum += 0.5 * el.pd * zv * zv + el.q * zv;
                const int k = e / NZ, i = e % NZ;
                if (k < N) { du[k * SR + NX + i] = yu; du[k * SR + NX + NZ + i] = yl; }
                else { du[N * SR + i] = yu; du[N * SR + NX + i] = yl; }
                if (e < NX && a.pin_dual) a.pin_dual[(size_t)b * NX + e] = -(el.pd * zv + el.q + gv);
            }
            const double *nus = polished ? NUP : NUA;
            for (int o = lane; o < N * NX; o += 64) du[(o / NX) * SR + (o % NX)] = nus[o];
            if (polished && a.stat_slot == 0 && a.snap_use == 0) {     // the call's first QP: its set is where the next call's first QP starts
                for (int e = lane; e < n; e += 64) ACT1[e] = ACT[e];
                s.act1_ok = (double)((((int)s.act1_ok | 1)) & ~(255 << 2));       // written now: age 0
            }
            if (polished && a.stat_slot == 1) {                        // the call's last QP: where the next call's last QP may start (as_warm_last)
                for (int e = lane; e < n; e += 64) ACT2[e] = ACT[e];
                s.act1_ok = (double)((((int)s.act1_ok | 2)) & ~(255 << 10));
            }
        }
        if (polished) s.fact_call = call_id;
        csum = wla::wave_sum(csum);
        nact = wla::wave_sum(nact);
        if (lane == 0) {
            if (a.qpstat) {
                int *qs = a.qpstat + ((size_t)b * 2 + a.stat_slot) * 8;
                qs[0] = it; qs[1] = (int)stp->ticks; qs[2] = (int)stp->fticks; qs[3] = (int)nact; qs[4] = (int)s.path / 10;   /* 0 cold, 1 warm (first QP: the previous call's first set; last QP: this call's first set), 2 last QP from the previous call's last set */
                qs[5] = (int)s.pol_round; qs[6] = status; qs[7] = (int)s.path % 10;   // 0: first attempt succeeded (warm set, or empty set on a cold solve); 1: cold interior point; 2: interior point restarted from an iterate copy; 3: active set from the empty set after a failed warm attempt
            }
            if (ok) a.cost[b] = csum;
            a.status[b] = status;
            a.iters[b] = it;
            double *kk = a.kkt + (size_t)b * 8;
            kk[0] = s.kst; kk[1] = s.kbox; kk[2] = s.ksign; kk[3] = s.mu; kk[4] = s.pst; kk[5] = s.pbox; kk[6] = stp->fticks; kk[7] = stp->ticks;   // [6],[7]: factor sweeps / block-tridiagonal solves this instance used
        }
    }
    if (lane == 0) { s.phase = phase; s.it = it; s.status = status; s.ticks = stp->ticks; s.fticks = stp->fticks; *stp = s; }
}

// The whole QP solve of one instance in one launch: the wave loops over its ticks (forward sweep, backward sweep, phase logic) until its
// instance is done, so instances advance independently -- no launch per tick, no host poll, no batch-wide barrier between ticks; the
// hardware's workgroup dispatcher fills the slots that finished instances free.  (Round 1 launched one kernel per part and tick and polled a
// counter of unfinished instances from the host.)
#ifndef QP_PERSIST_WAVES_PER_SIMD
#define QP_PERSIST_WAVES_PER_SIMD 3
#endif

template <int NX, int NU, bool MX = false>     // MX: the mixed-precision sweeps (ne_forward_mx / ne_backward_mx, section 2.4 of DESIGN.md), same loop and phase logic
// resume != 0: the solve was suspended by an earlier launch at its deadline (state in HBM: QpState, workspace) and continues where it stopped.
// deadline: wall-clock tick (100 MHz) after which the solve suspends itself between two block solves (~0: never).  Returns 1 when the solve has
// ended (or took no part), 0 when it was suspended.
// fin_count / cut_count (slsqp_cl_run): the solve also suspends itself once cut_count chains of its launch have ended (the few solves still running
// then are the ones that would keep the launch alive on their own).
__device__ __forceinline__ int qp_solve_dev(const QpArgs &a, int b, int lane, double *sm, int max_ticks, int resume = 0, unsigned long long deadline = ~0ULL,
                                            const unsigned *fin_count = nullptr, unsigned cut_count = 0xFFFFFFFFu) {
    if (!resume) {
        if (a.run && !a.run[b]) {       // not part of this solve: its statistics slot says so (status -1)
            if (a.qpstat && lane < 8) a.qpstat[((size_t)b * 2 + a.stat_slot) * 8 + lane] = (lane == 6) ? -1 : 0;
            return 1;
        }
        phase_update<NX, NU>(a, 1, b, lane);
        wla::wsync_mem();
    }
    int finished = 1;
    unsigned long long n_sweeps = 0, n_factor = 0, n_fstages = 0, n_bwd_skipped = 0;
#ifdef QP_STAMP
    long long c_fwd = 0, c_bwd = 0, c_ph = 0, c_t0 = __builtin_readcyclecounter(), c_fwdf = 0;
#define QSTAMP(acc) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long t_ = __builtin_readcyclecounter(); acc += t_ - c_last; c_last = t_; } while (0)
    long long c_last = c_t0;
#else
#define QSTAMP(acc) do {} while (0)
#endif
    for (int t = 0; t < max_ticks; t++) {
        // The instance index and the lane id are laundered through empty asm statements at the head of every part: nothing computed from them
        // is loop-invariant for the compiler then, so it cannot hoist the phase logic's per-element loads (bounds, weights, linear cost) out of
        // the tick loop and keep them in registers across the sweeps (256 VGPRs + 118 spilled when it does; 168-194 like this).
        LAUNDER_B(b);
        asm volatile("" : "+v"(lane));
        QpState *st = (QpState *)a.state + b;
        const int phase = (int)st->phase;
        if (phase == P_DONE) break;
        {   // suspend? (wave-uniform: a scalar clock read, one counter read broadcast from the first lane)
            int stop = (deadline != ~0ULL && wall_clock64() > deadline) ? 1 : 0;
            if (fin_count) stop |= (__hip_atomic_load(fin_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cut_count) ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(stop)) { finished = 0; break; }
        }
        const FwdPlan fp = fwd_plan(st, phase, a.N);
        const bool factor = fp.factor;
        double bmax = 0.0;
        int f;
        if constexpr (MX) f = ne_forward_mx<NX, NU>(sm, make_neg<NX, NU>(a, b), fp.factor, fp.eflag, fp.delta, lane, &bmax, fp.k0);
        else f = ne_forward<NX, NU>(sm, make_neg<NX, NU>(a, b), fp.factor, fp.eflag, fp.delta, lane, &bmax, fp.k0, fp.ks);
        if (lane == 0) {
            if (phase == P_POL1 || phase == P_POL2) st->pbox = bmax;
            st->ticks += 1.0;
            if (factor) { st->fticks += 1.0; if (phase == P_POL0) st->pol_fail = f; }
        }
        n_sweeps++; n_factor += factor ? 1 : 0; n_fstages += factor ? (unsigned long long)(a.N - fp.k0) : 0ULL;
        // (wave-uniform) the sweep only had to measure the equality residual of a solve that is otherwise certified, and it is tiny: no backward
        // sweep, phase_update accepts the un-refined solve (same test there)
        const bool res_only_done = !MX && phase == P_POL2 && st->res_only != 0.0 && bmax < RES_ONLY_TOL * st->qscale;
        n_bwd_skipped += res_only_done ? 1 : 0;
        wla::wsync_mem();
#ifdef QP_STAMP
        if (factor) QSTAMP(c_fwdf); else QSTAMP(c_fwd);
#endif
        LAUNDER_B(b);
        asm volatile("" : "+v"(lane));
        if (!res_only_done) {
            if constexpr (MX) ne_backward_mx<NX, NU>(sm, make_neg<NX, NU>(a, b), lane);
            else ne_backward<NX, NU>(sm, make_neg<NX, NU>(a, b), lane);
        }
        wla::wsync_mem();
        QSTAMP(c_bwd);
        LAUNDER_B(b);
        asm volatile("" : "+v"(lane));
        phase_update<NX, NU>(a, 0, b, lane, sm);
        wla::wsync_mem();
        QSTAMP(c_ph);
    }
#ifdef QP_STAMP
    if (lane == 0) { double *kk = a.kkt + (size_t)b * 8; kk[2] = (double)c_fwdf; kk[3] = (double)(double)n_fstages; kk[4] = (double)c_fwd; kk[5] = (double)c_bwd; kk[6] = (double)c_ph; kk[7] = (double)(__builtin_readcyclecounter() - c_t0); }
#endif
    // (a solve counts as one that ran when it ENDS with at least one block solve behind it, whatever number of launches it was spread over)
    if (lane == 0) { atomicAdd(a.inst_launches, n_sweeps); atomicAdd(a.inst_launches + 1, n_factor); atomicAdd(a.inst_launches + 2, n_fstages); atomicAdd(a.inst_launches + 3, (finished && ((QpState *)a.state + b)->ticks > 0.0) ? 1ULL : 0ULL); if (n_bwd_skipped) atomicAdd(a.inst_launches + 4, n_bwd_skipped); }
    return finished;
}
template <int NX, int NU, bool MX = false>
__global__ __launch_bounds__(64, QP_PERSIST_WAVES_PER_SIMD) void k_qp_solve(QpArgs a, int max_ticks) {
    const int b = blockIdx.x;
    if (b >= a.B) return;
    extern __shared__ double sm[];
    qp_solve_dev<NX, NU, MX>(a, b, threadIdx.x, sm, max_ticks);
}

// ------------------------------------------------------------------------------------------------
// bounds of the un-tightened QP: QP.update_dynamics (qp_jit.py:268-273) + offset_constraints (:595-610)
// ------------------------------------------------------------------------------------------------
struct BoundsArgs { int B, N, NX, NI, NIF; const double *g, *gN, *c; double *ubg, *lbg; double eps; const int *run; };
__device__ __forceinline__ void set_bounds_row(const BoundsArgs &a, int b, int r) {
    const int SR = a.NX + a.NI, mb = a.N * SR + a.NIF;
    double u, l;
    if (r < a.N * SR) {
        const int k = r / SR, i = r % SR;
        if (i < a.NX) { const double cv = a.c[((size_t)b * a.N + k) * a.NX + i]; u = -cv + a.eps; l = -cv - a.eps; }
        else { u = a.g[((size_t)b * a.N + k) * a.NI + (i - a.NX)] + a.eps; l = -1e20; }
    } else { u = a.gN[(size_t)b * a.NIF + (r - a.N * SR)] + a.eps; l = -1e20; }
    a.ubg[(size_t)b * mb + r] = u; a.lbg[(size_t)b * mb + r] = l;
}
__global__ void k_set_bounds(BoundsArgs a) {
    const int SR = a.NX + a.NI, mb = a.N * SR + a.NIF;
    const size_t tot = (size_t)a.B * mb;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = idx / mb, r = idx % mb;
        if (a.run && !a.run[b]) continue;
        set_bounds_row(a, b, r);
    }
}

// ------------------------------------------------------------------------------------------------
// evaluate_dual_eta (fast_SLS_jit.py:475-487)
// ------------------------------------------------------------------------------------------------
// stale[b] bits: 1 eta / eta_f and 2 K hold values from before the last slsqp_reset (zeroed on demand by slsqp_get); 8 beta / beta_f hold a
// sweep's values (not the eps of initialize_backoff); 16 eta / eta_f hold only column 0 of a first fast-SLS iteration (broadcast on demand);
// 32 K is still in its compact form Kc (shared Riccati recursion; broadcast on demand)
struct EtaArgs { int B, N, NX, NI, NIF; const double *dual, *beta, *beta_f; const int *run; double *eta, *eta_f; double eps; int *stale; int first_iter; };
// (evaluate_dual_eta runs inside k_after_qp, slsqp_api.hip: first iteration of a solve -- beta = eps in every entry, so eta[k,j] = mu_k / (2 sqrt(eps))
// for every j <= k and only column 0 is written, which is all the shared Riccati recursion reads; slsqp_get broadcasts it when the array is asked for)
// eta[k,j] = eta[k,0] (1 <= j <= k), eta_f[j] = eta_f[0] for the instances whose arrays hold only column 0
__global__ void k_eta_broadcast(int N, int NI, int NIF, int *stale, double *eta, double *eta_f) {
    const int b = blockIdx.x;
    if (!(stale[b] & 16)) return;
    double *et = eta + (size_t)b * N * N * NI, *ef = eta_f + (size_t)b * (N + 1) * NIF;
    for (int o = threadIdx.x; o < N * N * NI; o += blockDim.x) {
        const int i = o % NI, j = (o / NI) % N, k = o / (NI * N);
        if (j >= 1 && j <= k) et[o] = et[((size_t)k * N) * NI + i];
    }
    for (int o = threadIdx.x; o < N * NIF; o += blockDim.x) ef[NIF + o] = ef[o % NIF];
    __syncthreads();
    if (threadIdx.x == 0) stale[b] &= ~16;
}
// K[k,j] = Kc[k] (j <= k) for the instances whose K array was not written by the shared sweep (stale bit 32)
__global__ void k_K_broadcast(int N, int NUNX, int *stale, const double *Kc, double *K) {
    const int b = blockIdx.x;
    if (!(stale[b] & 32)) return;
    const double *kc = Kc + (size_t)b * N * NUNX;
    double *kg = K + (size_t)b * N * (N + 1) * NUNX;
    for (int o = threadIdx.x; o < N * (N + 1) * NUNX; o += blockDim.x) {
        const int e = o % NUNX, j = (o / NUNX) % (N + 1), k = o / (NUNX * (N + 1));
        kg[o] = (j <= k) ? kc[(size_t)k * NUNX + e] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) stale[b] &= ~32;
}
// ------------------------------------------------------------------------------------------------
// check_convergence_socp (fast_SLS_jit.py:581-600): state persists across calls
// ------------------------------------------------------------------------------------------------
struct ConvArgs { int B, n; const double *primal; double *prev; int *has_prev; const int *run; int *conv; double tol; };
// (runs inside k_after_qp, slsqp_api.hip)

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
    double *ct_part;           // (B,N+1) per-column part of cost_tube^2 = || blkdiag(Q_reg..,Q_reg_f,R_reg..) [Phi_x;Phi_u] ||_F^2 (util/SLS.py:38-46)
    double eps;
};

template <int NX, int NU>
__host__ __device__ constexpr int sweep_lds_doubles() {
    // sA sS sYm (3 NX^2: A + B K is formed in place in sA, y (A + B K) in sS, the propagate phase reuses sS / sYm for Phi, NW == NX)
    // + sB sX sF sK sPu (5 NX NU) + sH (NU^2) + sC (NX+NU)
    return 3 * NX * NX + 5 * NX * NU + NU * NU + (NX + NU) + 8;
}

// SWEEP_MFMA (default 1): the dense products of the sweep (A'S, y Acl, Acl Phi, B'S, x A, K Phi, A + B K) run on the fp64 matrix core for
// NX >= 13 (wla::gemm_mfma: one 16x16 tile, the 17th row / column / k on the vector ALU beside it); 0 = vector-ALU GEMMs only.
#ifndef SWEEP_MFMA
#define SWEEP_MFMA 1
#endif
#ifndef SWEEP_WAVES_PER_SIMD
#define SWEEP_WAVES_PER_SIMD 3
#endif
template <int NX, int NU>
__global__ __launch_bounds__(64, SWEEP_WAVES_PER_SIMD) void k_sweep(SweepArgs a) {
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
    double *sA = p; p += NX * NX; double *sS = p; p += NX * NX; double *sYm = p; p += NX * NX;
    double *sAcl = sA;                     // A + B K replaces A_k in place (every use of A_k precedes it)
    double *sSn = sS;                      // y (A + B K) lands where S_{k+1} was (dead once B'S and A'S are formed), symmetrised in place
    double *sPhi = sS, *sPhi2 = sYm;       // the Riccati buffers are dead in the propagate phase
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
    // software pipeline: next stage's A_k, B_k, eta are fetched into registers while the current stage is processed
    constexpr int RA = (NX * NX + 63) / 64, RB = (NX * NU + 63) / 64;
    double rA[RA], rB[RB], rC = 0.0;
    const int lz = min(lane, NZ - 1);
    const double regd = (lz < NX) ? a.cst.Qregd[lz] : a.cst.Rregd[lz - NX];
    auto fetch = [&](int k) {
        const double *Ak = gA + (size_t)k * NX * NX, *Bk = gB + (size_t)k * NX * NU;
#pragma unroll
        for (int r = 0; r < RA; r++) rA[r] = Ak[min(r * 64 + lane, NX * NX - 1)];
#pragma unroll
        for (int r = 0; r < RB; r++) rB[r] = Bk[min(r * 64 + lane, NX * NU - 1)];
        const double *e = eta + ((size_t)k * N + j) * NI;
        rC = e[lz] + e[NZ + lz] + regd;
    };
    if (N - 1 >= j) fetch(N - 1);
    for (int k = N - 1; k >= j; k--) {
#pragma unroll
        for (int r = 0; r < RA; r++) { const int o = r * 64 + lane; if (o < NX * NX) sA[o] = rA[r]; }
#pragma unroll
        for (int r = 0; r < RB; r++) { const int o = r * 64 + lane; if (o < NX * NU) sB[o] = rB[r]; }
        if (lane < NZ) sC[lane] = rC;
        wla::wsync();
        if (k - 1 >= j) fetch(k - 1);
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NU, NX, NX, true, false>(sB, NU, sS, NX, sX, NX, lane);
        else
#endif
        wla::gemm_blk<NU, NX, NX, true, false, 1, 2, false>(sB, NU, sS, NX, sX, NX, 1.0, lane);   // x = B' S   (NU x NX)
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NX, NX, NX, true, false>(sA, NX, sS, NX, sYm, NX, lane);   // y = A' S on the fp64 matrix core
        else
#endif
        wla::gemm_blk<NX, NX, NX, true, false, 3, 2, false>(sA, NX, sS, NX, sYm, NX, 1.0, lane);  // y = A' S   (NX x NX)
        wla::wsync();
        {   // H = x B (NU x NU), the sum over k split over three lane groups (lane = entry + NU^2 g)
            constexpr int NH = NU * NU;
            const int g3 = lane / NH, o = lane - g3 * NH, hi = o / NU, hj = o % NU;
            double hs = 0.0;
            if (g3 < 3) {
#pragma unroll
                for (int q = 0; q < (NX + 2) / 3; q++) { const int kk = 3 * q + g3; if (kk < NX) hs = fma(sX[hi * NX + kk], sB[kk * NU + hj], hs); }
            }
            hs = hs + __shfl(hs, lane + NH) + __shfl(hs, lane + 2 * NH);
            if (lane < NH) sH[lane] = hs;
        }
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NU, NX, NX, false, false>(sX, NX, sA, NX, sF, NX, lane);
        else
#endif
        wla::gemm_blk<NU, NX, NX, false, false, 1, 2, false>(sX, NX, sA, NX, sF, NX, 1.0, lane);  // F = x A
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
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NX, NX, NU, false, false, true>(sB, NU, sK, NX, sAcl, NX, lane, sA, NX);
        else
#endif
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) {
            const int i = o / NX, jj = o % NX;
            double s = sA[o];
#pragma unroll
            for (int u = 0; u < NU; u++) s = fma(sB[i * NU + u], sK[u * NX + jj], s);
            sAcl[o] = s;
        }
        wla::wsync();
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NX, NX, NX, false, false>(sYm, NX, sAcl, NX, sSn, NX, lane);
        else
#endif
        wla::gemm_blk<NX, NX, NX, false, false, 3, 2, false>(sYm, NX, sAcl, NX, sSn, NX, 1.0, lane);  // y (A + B K)
        wla::wsync();
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) {   // S = (Sn + Sn')/2 + diag, in place: the lane of (i,j), i >= j, writes both (i,j) and (j,i)
            const int i = o / NX, jj = o % NX;
            if (i >= jj) {
                const double v = 0.5 * (sSn[o] + sSn[jj * NX + i]) + ((i == jj) ? sC[i] : 0.0);
                sS[o] = v; sS[jj * NX + i] = v;
            }
        }
        wla::wsync();
    }
    wla::wsync_mem();   // K written above is re-read below by other lanes of this wave
    // propagate column j and accumulate row norms
    const double *Eg = a.E + (a.E_per_instance ? (size_t)b * (N + 1) * NX * NW : 0) + (size_t)j * NX * NW;
#pragma unroll
    for (int o = lane; o < NX * NW; o += 64) sPhi[o] = Eg[o];
    wla::wsync();
    double *Pc = sPhi, *Pn = sPhi2;
    double rK[RB];
    double ctube = 0.0;
    auto fetch2 = [&](int k) {
        const double *Ak = gA + (size_t)k * NX * NX, *Bk = gB + (size_t)k * NX * NU;
        const double *Kg = gK + ((size_t)k * (N + 1) + j) * NU * NX;
#pragma unroll
        for (int r = 0; r < RA; r++) rA[r] = Ak[min(r * 64 + lane, NX * NX - 1)];
#pragma unroll
        for (int r = 0; r < RB; r++) { const int o = min(r * 64 + lane, NX * NU - 1); rB[r] = Bk[o]; rK[r] = Kg[o]; }
    };
    if (j < N) fetch2(j);
    for (int k = j; k < N; k++) {
#pragma unroll
        for (int r = 0; r < RA; r++) { const int o = r * 64 + lane; if (o < NX * NX) sA[o] = rA[r]; }
#pragma unroll
        for (int r = 0; r < RB; r++) { const int o = r * 64 + lane; if (o < NX * NU) { sB[o] = rB[r]; sK[o] = rK[r]; } }
        wla::wsync();
        if (k + 1 < N) fetch2(k + 1);
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NU, NW, NX, false, false>(sK, NX, Pc, NW, sPu, NW, lane);
        else
#endif
        wla::gemm_blk<NU, NW, NX, false, false, 1, 2, false>(sK, NX, Pc, NW, sPu, NW, 1.0, lane);  // Phi_u = K Phi_x
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NX, NX, NU, false, false, true>(sB, NU, sK, NX, sAcl, NX, lane, sA, NX);
        else
#endif
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
        {   // row norms, the sum over w split over three lane groups (lane = row + NZ g)
            const int g3 = lane / NZ, rw = lane - g3 * NZ;
            const double *row = (rw < NX) ? Pc + rw * NW : sPu + (rw - NX) * NW;
            double s = 0.0;
            if (g3 < 3) {
#pragma unroll
                for (int q = 0; q < (NW + 2) / 3; q++) { const int w = 3 * q + g3; if (w < NW) s = fma(row[w], row[w], s); }
            }
            s = s + __shfl(s, lane + NZ) + __shfl(s, lane + 2 * NZ);
            if (lane < NZ) {
                const double wr = (lane < NX) ? a.cst.Qregd[lane] : a.cst.Rregd[lane - NX];
                ctube = fma(wr * wr, s, ctube);       // weighted row norms: this column's share of cost_tube^2
                s = fmax(s, a.eps);
                double *bo = beta + ((size_t)k * N + j) * NI;
                bo[lane] = s; bo[NZ + lane] = s;
            }
        }
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NX, NW, NX, false, false>(sAcl, NX, Pc, NW, Pn, NW, lane);        // Phi_{k+1} = Acl Phi_k
        else
#endif
        wla::gemm_blk<NX, NW, NX, false, false, 3, 2, false>(sAcl, NX, Pc, NW, Pn, NW, 1.0, lane);
        wla::wsync();
        double *t = Pc; Pc = Pn; Pn = t;
    }
    if (lane < NX) {
        const double *row = Pc + lane * NW;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w++) s = fma(row[w], row[w], s);
        ctube = fma(a.cst.Qregfd[lane] * a.cst.Qregfd[lane], s, ctube);
        s = fmax(s, a.eps);
        beta_f[j * NIF + lane] = s; beta_f[j * NIF + NX + lane] = s;
    }
    ctube = wla::wave_sum(ctube);
    if (lane == 0 && a.ct_part) a.ct_part[(size_t)b * (N + 1) + j] = ctube;
}

// ------------------------------------------------------------------------------------------------
// SLS sweep for a general constraint matrix G (ni x (nx+nu)), Gf (ni_f x nx) -- what the reference's kernels take
// (fast_SLS_jit.py:76-79: C = G' diag(eta) G, its x-x and u-u blocks, cross block ignored; :138-170: beta_i = || G_x,i Phi_x + G_u,i Phi_u ||^2;
// Pendulum.replace_constraints, dyn/pendulum.py:146, builds such a model).  Same mapping as k_sweep (one wave per (instance, column)); the
// constraint-dependent parts run on the vector ALU from G in global memory -- this is the sweep-level boundary's path (slsqp_sweep), not the
// closed loop's, where every plant of the reference has G = [I;-I].
// ------------------------------------------------------------------------------------------------
struct SweepGenArgs { SweepArgs s; const double *G, *Gf; int NI, NIF; };
template <int NX, int NU>
__host__ __device__ constexpr int sweep_gen_lds_doubles() { return 4 * NX * NX + 5 * NX * NU + 2 * NU * NU + 8; }
template <int NX, int NU>
__global__ __launch_bounds__(64) void k_sweep_gen(SweepGenArgs ga) {
    const SweepArgs &a = ga.s;
    constexpr int NZ = NX + NU, NW = NX;
    const int N = a.N, lane = threadIdx.x, NI = ga.NI, NIF = ga.NIF;
    const int ncol = N + 1, b = blockIdx.x / ncol, j = blockIdx.x % ncol;
    if (b >= a.B || (a.run && !a.run[b])) return;
    extern __shared__ double sm[];
    double *p = sm;
    double *sA = p; p += NX * NX; double *sS = p; p += NX * NX; double *sYm = p; p += NX * NX; double *sCx = p; p += NX * NX;
    double *sAcl = sA, *sSn = sS, *sPhi = sS, *sPhi2 = sYm;
    double *sB = p; p += NX * NU; double *sX = p; p += NX * NU; double *sF = p; p += NX * NU; double *sK = p; p += NX * NU;
    double *sPu = p; p += NU * NW; double *sH = p; p += NU * NU; double *sCu = p; p += NU * NU;
    const double *gA = a.A + (size_t)b * N * NX * NX, *gB = a.Bm + (size_t)b * N * NX * NU;
    const double *eta = a.eta + (size_t)b * N * N * NI, *eta_f = a.eta_f + (size_t)b * (N + 1) * NIF;
    double *gK = a.K + (size_t)b * N * (N + 1) * NU * NX;
    double *beta = a.beta + (size_t)b * N * N * NI, *beta_f = a.beta_f + (size_t)b * (N + 1) * NIF;
    const double *G = ga.G, *Gf = ga.Gf;
    // terminal: S[N,j] = Gf' diag(eta_f[j]) Gf + Q_reg_f
    for (int o = lane; o < NX * NX; o += 64) {
        const int i = o / NX, jj = o % NX;
        double s = (i == jj) ? a.cst.Qregfd[i] : 0.0;
        for (int r = 0; r < NIF; r++) s = fma(Gf[r * NX + i] * eta_f[j * NIF + r], Gf[r * NX + jj], s);
        sS[o] = s;
    }
    wla::wsync();
    for (int k = N - 1; k >= j; k--) {
        const double *Ak = gA + (size_t)k * NX * NX, *Bk = gB + (size_t)k * NX * NU, *e = eta + ((size_t)k * N + j) * NI;
        for (int o = lane; o < NX * NX; o += 64) {
            const int i = o / NX, jj = o % NX;
            sA[o] = Ak[o];
            double s = (i == jj) ? a.cst.Qregd[i] : 0.0;                                   // C_xx + Q_reg
            for (int r = 0; r < NI; r++) s = fma(G[r * NZ + i] * e[r], G[r * NZ + jj], s);
            sCx[o] = s;
        }
        for (int o = lane; o < NX * NU; o += 64) sB[o] = Bk[o];
        for (int o = lane; o < NU * NU; o += 64) {
            const int i = o / NU, jj = o % NU;
            double s = (i == jj) ? a.cst.Rregd[i] : 0.0;                                   // C_uu + R_reg
            for (int r = 0; r < NI; r++) s = fma(G[r * NZ + NX + i] * e[r], G[r * NZ + NX + jj], s);
            sCu[o] = s;
        }
        wla::wsync();
        wla::gemm_blk<NU, NX, NX, true, false, 1, 2, false>(sB, NU, sS, NX, sX, NX, 1.0, lane);   // x = B' S
        wla::gemm_blk<NX, NX, NX, true, false, 3, 2, false>(sA, NX, sS, NX, sYm, NX, 1.0, lane);  // y = A' S
        wla::wsync();
        for (int o = lane; o < NU * NU; o += 64) {                                                // H = C_uu + R_reg + x B
            const int i = o / NU, jj = o % NU;
            double s = sCu[o];
            for (int kk = 0; kk < NX; kk++) s = fma(sX[i * NX + kk], sB[kk * NU + jj], s);
            sH[o] = s;
        }
        wla::gemm_blk<NU, NX, NX, false, false, 1, 2, false>(sX, NX, sA, NX, sF, NX, 1.0, lane);  // F = x A
        wla::wsync();
        if (lane < NX) {   // K = -H^{-1} F, one column per lane (H symmetric positive definite: C_uu is a Gram matrix, R_reg > 0, x B = B' S B)
            double f[NU];
#pragma unroll
            for (int u = 0; u < NU; u++) f[u] = sF[u * NX + lane];
            wla::spd_solve_small<NU>(sH, NU, f);
#pragma unroll
            for (int u = 0; u < NU; u++) sK[u * NX + lane] = -f[u];
        }
        wla::wsync();
        double *Kg = gK + ((size_t)k * (N + 1) + j) * NU * NX;
        for (int o = lane; o < NU * NX; o += 64) Kg[o] = sK[o];
        for (int o = lane; o < NX * NX; o += 64) {                                                // Acl = A + B K
            const int i = o / NX, jj = o % NX;
            double s = sA[o];
#pragma unroll
            for (int u = 0; u < NU; u++) s = fma(sB[i * NU + u], sK[u * NX + jj], s);
            sAcl[o] = s;
        }
        wla::wsync();
        wla::gemm_blk<NX, NX, NX, false, false, 3, 2, false>(sYm, NX, sAcl, NX, sSn, NX, 1.0, lane);  // y (A + B K)
        wla::wsync();
        for (int o = lane; o < NX * NX; o += 64) {   // S = C_xx + Q_reg + sym(y Acl), in place
            const int i = o / NX, jj = o % NX;
            if (i >= jj) {
                const double v = 0.5 * (sSn[o] + sSn[jj * NX + i]) + 0.5 * (sCx[o] + sCx[jj * NX + i]);
                sS[o] = v; sS[jj * NX + i] = v;
            }
        }
        wla::wsync();
    }
    wla::wsync_mem();
    // propagate column j, beta from the rows of G [Phi_x; Phi_u]
    const double *Eg = a.E + (a.E_per_instance ? (size_t)b * (N + 1) * NX * NW : 0) + (size_t)j * NX * NW;
    for (int o = lane; o < NX * NW; o += 64) sPhi[o] = Eg[o];
    wla::wsync();
    double *Pc = sPhi, *Pn = sPhi2;
    double ctube = 0.0;
    for (int k = j; k < N; k++) {
        const double *Ak = gA + (size_t)k * NX * NX, *Bk = gB + (size_t)k * NX * NU, *Kg = gK + ((size_t)k * (N + 1) + j) * NU * NX;
        for (int o = lane; o < NX * NX; o += 64) sA[o] = Ak[o];
        for (int o = lane; o < NX * NU; o += 64) { sB[o] = Bk[o]; sK[o] = Kg[o]; }
        wla::wsync();
        wla::gemm_blk<NU, NW, NX, false, false, 1, 2, false>(sK, NX, Pc, NW, sPu, NW, 1.0, lane);  // Phi_u = K Phi_x
        for (int o = lane; o < NX * NX; o += 64) {
            const int i = o / NX, jj = o % NX;
            double s = sA[o];
#pragma unroll
            for (int u = 0; u < NU; u++) s = fma(sB[i * NU + u], sK[u * NX + jj], s);
            sAcl[o] = s;
        }
        wla::wsync();
        for (int r = lane; r < NI; r += 64) {
            double s = 0.0;
            for (int w = 0; w < NW; w++) {
                double v = 0.0;
                for (int c = 0; c < NX; c++) v = fma(G[r * NZ + c], Pc[c * NW + w], v);
                for (int c = 0; c < NU; c++) v = fma(G[r * NZ + NX + c], sPu[c * NW + w], v);
                s = fma(v, v, s);
            }
            beta[((size_t)k * N + j) * NI + r] = fmax(s, a.eps);
        }
        if (lane < NZ) {   // this column's share of cost_tube^2 (weighted row norms of [Phi_x; Phi_u])
            const double *row = (lane < NX) ? Pc + lane * NW : sPu + (lane - NX) * NW;
            double s = 0.0;
            for (int w = 0; w < NW; w++) s = fma(row[w], row[w], s);
            const double wr = (lane < NX) ? a.cst.Qregd[lane] : a.cst.Rregd[lane - NX];
            ctube = fma(wr * wr, s, ctube);
        }
        wla::gemm_blk<NX, NW, NX, false, false, 3, 2, false>(sAcl, NX, Pc, NW, Pn, NW, 1.0, lane);    // Phi_{k+1} = Acl Phi_k
        wla::wsync();
        double *t = Pc; Pc = Pn; Pn = t;
    }
    for (int r = lane; r < NIF; r += 64) {
        double s = 0.0;
        for (int w = 0; w < NW; w++) {
            double v = 0.0;
            for (int c = 0; c < NX; c++) v = fma(Gf[r * NX + c], Pc[c * NW + w], v);
            s = fma(v, v, s);
        }
        beta_f[j * NIF + r] = fmax(s, a.eps);
    }
    if (lane < NX) {
        const double *row = Pc + lane * NW;
        double s = 0.0;
        for (int w = 0; w < NW; w++) s = fma(row[w], row[w], s);
        ctube = fma(a.cst.Qregfd[lane] * a.cst.Qregfd[lane], s, ctube);
    }
    ctube = wla::wave_sum(ctube);
    if (lane == 0 && a.ct_part) a.ct_part[(size_t)b * (N + 1) + j] = ctube;
}

// ------------------------------------------------------------------------------------------------
// The sweep of the FIRST fast-SLS iteration of a solve.  initialize_backoff (fast_SLS_jit.py:444-454) has just reset every beta[k,j] to eps, so
// evaluate_dual_eta (:475-487) gives eta[k,j] = mu_k / (2 sqrt(eps)) for every column j <= k and eta_f[j] = mu_f / (2 sqrt(eps)) for every j: the
// cost blocks of the N+1 Riccati recursions of _backward_solve_numba (:65-84) are the same, hence S[k,j] = S[k], K[k,j] = K[k] and
// A_k + B_k K_k for all j <= k.  One wave per instance runs that recursion once (k_sweep_ric1: 20 stages instead of 210) and leaves K_k and
// A_k + B_k K_k in a compact scratch; one wave per (instance, column) then only propagates Phi and takes the row norms (k_sweep_prop).  Same
// arithmetic in the same order as k_sweep, which remains for the later iterations of a solve (beta, hence eta, then differ from column to column).
// In the rocket script's setting (one fast-SLS step per MPC step) every sweep is a first one.
// ------------------------------------------------------------------------------------------------
struct SweepSharedArgs {
    SweepArgs s;
    double *Kc, *Aclc;         // scratch (B,N,NU,NX), (B,N,NX,NX)
    int *stale;                // bit 32 set by k_sweep_ric1: K[k,j] = Kc[k] (j <= k) not written out yet (slsqp_get broadcasts it on demand)
};
template <int NX, int NU>
__device__ __forceinline__ void sweep_ric1_dev(const SweepSharedArgs &aa, int b, int lane, double *sm) {
    const SweepArgs &a = aa.s;
    using L = Lay<NX, NU>;
    constexpr int NZ = L::NZ, NI = L::NI, NIF = L::NIF;
    const int N = a.N, j = 0;
    double *p = sm;
    double *sA = p; p += NX * NX; double *sS = p; p += NX * NX; double *sYm = p; p += NX * NX;
    double *sAcl = sA, *sSn = sS;
    double *sB = p; p += NX * NU; double *sX = p; p += NX * NU; double *sF = p; p += NX * NU; double *sK = p; p += NX * NU;
    p += NU * NX; double *sH = p; p += NU * NU; double *sC = p; p += NZ;
    const double *gA = a.A + (size_t)b * N * NX * NX, *gB = a.Bm + (size_t)b * N * NX * NU;
    const double *eta = a.eta + (size_t)b * N * N * NI, *eta_f = a.eta_f + (size_t)b * (N + 1) * NIF;
    double *gKc = aa.Kc + (size_t)b * N * NU * NX, *gAc = aa.Aclc + (size_t)b * N * NX * NX;
#pragma unroll
    for (int o = lane; o < NX * NX; o += 64) {
        const int i = o / NX, jj = o % NX;
        sS[o] = (i == jj) ? eta_f[j * NIF + i] + eta_f[j * NIF + NX + i] + a.cst.Qregfd[i] : 0.0;
    }
    wla::wsync();
    constexpr int RA = (NX * NX + 63) / 64, RB = (NX * NU + 63) / 64;
    double rA[RA], rB[RB], rC = 0.0;
    const int lz = min(lane, NZ - 1);
    const double regd = (lz < NX) ? a.cst.Qregd[lz] : a.cst.Rregd[lz - NX];
    auto fetch = [&](int k) {
        const double *Ak = gA + (size_t)k * NX * NX, *Bk = gB + (size_t)k * NX * NU;
#pragma unroll
        for (int r = 0; r < RA; r++) rA[r] = Ak[min(r * 64 + lane, NX * NX - 1)];
#pragma unroll
        for (int r = 0; r < RB; r++) rB[r] = Bk[min(r * 64 + lane, NX * NU - 1)];
        const double *e = eta + ((size_t)k * N + j) * NI;
        rC = e[lz] + e[NZ + lz] + regd;
    };
    if (lane == 0 && aa.stale) aa.stale[b] = (aa.stale[b] & ~2) | 32;
    fetch(N - 1);
    for (int k = N - 1; k >= 0; k--) {
#pragma unroll
        for (int r = 0; r < RA; r++) { const int o = r * 64 + lane; if (o < NX * NX) sA[o] = rA[r]; }
#pragma unroll
        for (int r = 0; r < RB; r++) { const int o = r * 64 + lane; if (o < NX * NU) sB[o] = rB[r]; }
        if (lane < NZ) sC[lane] = rC;
        wla::wsync();
        if (k - 1 >= 0) fetch(k - 1);
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NU, NX, NX, true, false>(sB, NU, sS, NX, sX, NX, lane);
        else
#endif
        wla::gemm_blk<NU, NX, NX, true, false, 1, 2, false>(sB, NU, sS, NX, sX, NX, 1.0, lane);   // x = B' S   (NU x NX)
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NX, NX, NX, true, false>(sA, NX, sS, NX, sYm, NX, lane);
        else
#endif
        wla::gemm_blk<NX, NX, NX, true, false, 3, 2, false>(sA, NX, sS, NX, sYm, NX, 1.0, lane);  // y = A' S   (NX x NX)
        wla::wsync();
        {
            constexpr int NH = NU * NU;
            const int g3 = lane / NH, o = lane - g3 * NH, hi = o / NU, hj = o % NU;
            double hs = 0.0;
            if (g3 < 3) {
#pragma unroll
                for (int q = 0; q < (NX + 2) / 3; q++) { const int kk = 3 * q + g3; if (kk < NX) hs = fma(sX[hi * NX + kk], sB[kk * NU + hj], hs); }
            }
            hs = hs + __shfl(hs, lane + NH) + __shfl(hs, lane + 2 * NH);
            if (lane < NH) sH[lane] = hs;
        }
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NU, NX, NX, false, false>(sX, NX, sA, NX, sF, NX, lane);
        else
#endif
        wla::gemm_blk<NU, NX, NX, false, false, 1, 2, false>(sX, NX, sA, NX, sF, NX, 1.0, lane);  // F = x A
        wla::wsync();
        if (lane < NU) sH[lane * NU + lane] += sC[NX + lane];
        wla::wsync();
        if (lane < NX) {
            double f[NU];
#pragma unroll
            for (int u = 0; u < NU; u++) f[u] = sF[u * NX + lane];
            wla::spd_solve_small<NU>(sH, NU, f);
#pragma unroll
            for (int u = 0; u < NU; u++) sK[u * NX + lane] = -f[u];
        }
        wla::wsync();
        double *Kg = gKc + (size_t)k * NU * NX;
#pragma unroll
        for (int o = lane; o < NU * NX; o += 64) Kg[o] = sK[o];
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NX, NX, NU, false, false, true>(sB, NU, sK, NX, sAcl, NX, lane, sA, NX);
        else
#endif
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) {
            const int i = o / NX, jj = o % NX;
            double s = sA[o];
#pragma unroll
            for (int u = 0; u < NU; u++) s = fma(sB[i * NU + u], sK[u * NX + jj], s);
            sAcl[o] = s;
        }
        wla::wsync();
        double *Ag = gAc + (size_t)k * NX * NX;
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) Ag[o] = sAcl[o];
#if SWEEP_MFMA
        if constexpr (NX >= 13) wla::gemm_mfma<NX, NX, NX, false, false>(sYm, NX, sAcl, NX, sSn, NX, lane);
        else
#endif
        wla::gemm_blk<NX, NX, NX, false, false, 3, 2, false>(sYm, NX, sAcl, NX, sSn, NX, 1.0, lane);  // y (A + B K)
        wla::wsync();
#pragma unroll
        for (int o = lane; o < NX * NX; o += 64) {
            const int i = o / NX, jj = o % NX;
            if (i >= jj) {
                const double v = 0.5 * (sSn[o] + sSn[jj * NX + i]) + ((i == jj) ? sC[i] : 0.0);
                sS[o] = v; sS[jj * NX + i] = v;
            }
        }
        wla::wsync();
    }
}
template <int NX, int NU>
__global__ __launch_bounds__(64, SWEEP_WAVES_PER_SIMD) void k_sweep_ric1(SweepSharedArgs aa) {
    const int b = blockIdx.x;
    if (b >= aa.s.B) return;
    if (aa.s.run && !aa.s.run[b]) return;
    extern __shared__ double sm[];
    sweep_ric1_dev<NX, NU>(aa, b, threadIdx.x, sm);
}

template <int NX, int NU>
__host__ __device__ constexpr int sweep_prop_lds_doubles() { return 5 * NX * NX + 3 * NX * NU + 8; }

// Phi propagation of TWO disturbance columns j0, j0 + 1 (ncols = 1: only j0) of one instance by one wave: per stage K_k and A_k + B_k K_k go to LDS
// once and serve both columns, whose products are independent chains on the matrix core (the wave waits on dependent LDS -> MFMA -> LDS round trips;
// with two columns every such trip carries twice the work).  Column j0 + 1 starts one stage later.  Per column the arithmetic is that of a wave
// that propagates it alone.
template <int NX, int NU>
__device__ __forceinline__ void sweep_prop_dev(const SweepSharedArgs &aa, int b, int j0, int ncols, int lane, double *sm) {
    const SweepArgs &a = aa.s;
    using L = Lay<NX, NU>;
    constexpr int NZ = L::NZ, NI = L::NI, NIF = L::NIF, NW = NX;
    const int N = a.N;
    double *p = sm;
    double *sAcl = p; p += NX * NX;
    double *PcA = p; p += NX * NX; double *PnA = p; p += NX * NX;
    double *PcB = p; p += NX * NX; double *PnB = p; p += NX * NX;
    double *sK = p; p += NX * NU; double *sPuA = p; p += NU * NW; double *sPuB = p; p += NU * NW;
    const double *gKc = aa.Kc + (size_t)b * N * NU * NX, *gAc = aa.Aclc + (size_t)b * N * NX * NX;
    double *beta = a.beta + (size_t)b * N * N * NI, *beta_f = a.beta_f + (size_t)b * (N + 1) * NIF;
    const double *EgA = a.E + (a.E_per_instance ? (size_t)b * (N + 1) * NX * NW : 0) + (size_t)j0 * NX * NW;
    const bool two = ncols > 1;
#pragma unroll
    for (int o = lane; o < NX * NW; o += 64) { PcA[o] = EgA[o]; if (two) PcB[o] = EgA[NX * NW + o]; }
    wla::wsync();
    constexpr int RA = (NX * NX + 63) / 64, RB = (NX * NU + 63) / 64;
    double rA[RA], rK[RB];
    double ctA = 0.0, ctB = 0.0;
    auto fetch2 = [&](int k) {
        const double *Ak = gAc + (size_t)k * NX * NX, *Kk = gKc + (size_t)k * NU * NX;
#pragma unroll
        for (int r = 0; r < RA; r++) rA[r] = Ak[min(r * 64 + lane, NX * NX - 1)];
#pragma unroll
        for (int r = 0; r < RB; r++) rK[r] = Kk[min(r * 64 + lane, NX * NU - 1)];
    };
    // row norms of [Phi_x; Phi_u] of column j at stage k -> beta, tube cost
    auto norms = [&](const double *Pc, const double *sPu, int k, int j, double &ctube) {
        const int g3 = lane / NZ, rw = lane - g3 * NZ;
        const double *row = (rw < NX) ? Pc + rw * NW : sPu + (rw - NX) * NW;
        double s = 0.0;
        if (g3 < 3) {
#pragma unroll
            for (int q = 0; q < (NW + 2) / 3; q++) { const int w = 3 * q + g3; if (w < NW) s = fma(row[w], row[w], s); }
        }
        s = s + __shfl(s, lane + NZ) + __shfl(s, lane + 2 * NZ);
        if (lane < NZ) {
            const double wr = (lane < NX) ? a.cst.Qregd[lane] : a.cst.Rregd[lane - NX];
            ctube = fma(wr * wr, s, ctube);
            s = fmax(s, a.eps);
            double *bo = beta + ((size_t)k * N + j) * NI;
            bo[lane] = s; bo[NZ + lane] = s;
        }
    };
    constexpr bool PAIR = (SWEEP_MFMA != 0) && NX >= 13;
    if (j0 < N) fetch2(j0);
    for (int k = j0; k < N; k++) {
#pragma unroll
        for (int r = 0; r < RA; r++) { const int o = r * 64 + lane; if (o < NX * NX) sAcl[o] = rA[r]; }
#pragma unroll
        for (int r = 0; r < RB; r++) { const int o = r * 64 + lane; if (o < NX * NU) sK[o] = rK[r]; }
        wla::wsync();
        if (k + 1 < N) fetch2(k + 1);
        const bool bB = two && k > j0;      // column j0 + 1 has its first stage at k = j0 + 1
        // Phi_u = K Phi_x and Phi_{k+1} = (A + B K) Phi_k: on the matrix core both at once (independent MFMA chains, Phi_k's operand read once)
        if constexpr (PAIR) {
            wla::gemm_mfma_pair<NU, NX, NW, NX>(sK, NX, sAcl, NX, PcA, NW, sPuA, NW, PnA, NW, lane);
            if (bB) wla::gemm_mfma_pair<NU, NX, NW, NX>(sK, NX, sAcl, NX, PcB, NW, sPuB, NW, PnB, NW, lane);
        } else {
            wla::gemm_blk<NU, NW, NX, false, false, 1, 2, false>(sK, NX, PcA, NW, sPuA, NW, 1.0, lane);  // Phi_u = K Phi_x
            if (bB) wla::gemm_blk<NU, NW, NX, false, false, 1, 2, false>(sK, NX, PcB, NW, sPuB, NW, 1.0, lane);
        }
        wla::wsync();
        norms(PcA, sPuA, k, j0, ctA);
        if (bB) norms(PcB, sPuB, k, j0 + 1, ctB);
        if constexpr (!PAIR) {
            wla::gemm_blk<NX, NW, NX, false, false, 3, 2, false>(sAcl, NX, PcA, NW, PnA, NW, 1.0, lane);   // Phi_{k+1} = Acl Phi_k
            if (bB) wla::gemm_blk<NX, NW, NX, false, false, 3, 2, false>(sAcl, NX, PcB, NW, PnB, NW, 1.0, lane);
        }
        wla::wsync();
        { double *t = PcA; PcA = PnA; PnA = t; }
        if (bB) { double *t = PcB; PcB = PnB; PnB = t; }
    }
    auto terminal = [&](const double *Pc, int j, double ctube) {
        if (lane < NX) {
            const double *row = Pc + lane * NW;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w++) s = fma(row[w], row[w], s);
            ctube = fma(a.cst.Qregfd[lane] * a.cst.Qregfd[lane], s, ctube);
            s = fmax(s, a.eps);
            beta_f[j * NIF + lane] = s; beta_f[j * NIF + NX + lane] = s;
        }
        ctube = wla::wave_sum(ctube);
        if (lane == 0 && a.ct_part) a.ct_part[(size_t)b * (N + 1) + j] = ctube;
    };
    terminal(PcA, j0, ctA);
    if (two) terminal(PcB, j0 + 1, ctB);
}
template <int NX, int NU>
__global__ __launch_bounds__(64, 3) void k_sweep_prop(SweepSharedArgs aa) {
    const SweepArgs &a = aa.s;
    const int ncol = a.N + 1, npair = (ncol + 1) / 2;
    int b, jp;
    {   // XCD-aware mapping: blocks i and i+8 share an XCD; all column pairs of an instance stay on one XCD (its K_k, A_k + B_k K_k stay in that L2)
        const int bid = blockIdx.x;
        const int Bfull = (a.B / 8) * 8;
        if (bid < Bfull * npair) {
            const int xcd = bid % 8, slot = bid / 8;
            b = (slot / npair) * 8 + xcd; jp = slot % npair;
        } else {
            const int r = bid - Bfull * npair;
            b = Bfull + r / npair; jp = r % npair;
        }
    }
    if (b >= a.B) return;
    if (a.run && !a.run[b]) return;
    extern __shared__ double sm[];
    sweep_prop_dev<NX, NU>(aa, b, 2 * jp, min(2, ncol - 2 * jp), threadIdx.x, sm);
}

// ------------------------------------------------------------------------------------------------
// backoff sums + tightened bounds (fast_SLS_jit.py:173-186, 556-569) ; one workgroup per instance
// ------------------------------------------------------------------------------------------------
struct TightenArgs {
    int B, N, NX, NU, NI, NIF;   // NI, NIF: rows of G, Gf (2(nx+nu), 2nx for the box constraints of the reference's plants)
    const double *beta, *beta_f, *g, *gf_raw, *c;
    const int *run;
    double *backoff, *backoff_f, *backoff_x, *backoff_u, *ubg;
    int write_ubg;
    const double *ct_part; double *cost_tube;   // (B,N+1) -> (B): cost_tube = sqrt(sum over columns)
};
__device__ __forceinline__ void tighten_dev(const TightenArgs &a, int b, int tid, int nthr) {
    const int NX = a.NX, NU = a.NU, NZ = NX + NU, NI = a.NI, NIF = a.NIF, N = a.N, SR = NX + NI, mb = N * SR + NIF;
    const double *be = a.beta + (size_t)b * N * N * NI, *bf = a.beta_f + (size_t)b * (N + 1) * NIF;
    double *bo = a.backoff + (size_t)b * N * NI, *bof = a.backoff_f + (size_t)b * NIF;
    double *bx = a.backoff_x + (size_t)b * (N + 1) * NX, *bu = a.backoff_u + (size_t)b * N * NU;
    double *ub = a.ubg + (size_t)b * mb;
    for (int o = tid; o < N * NI; o += nthr) {
        const int k = o / NI, i = o % NI;
        double acc = 0.0;
        for (int j = 0; j <= k; j++) acc += sqrt(be[((size_t)k * N + j) * NI + i]);
        bo[o] = acc;
        if (i < NX) bx[k * NX + i] = acc;
        else if (i < NZ) bu[k * NU + (i - NX)] = acc;
        if (a.write_ubg) ub[k * SR + NX + i] = a.g[((size_t)b * N + k) * NI + i] - acc;   // no +eps (quirk q3)
    }
    for (int o = tid; o < NIF; o += nthr) {
        double acc = 0.0;
        for (int j = 0; j <= N; j++) acc += sqrt(bf[j * NIF + o]);
        bof[o] = acc;
        if (o < NX) bx[N * NX + o] = acc;
        if (a.write_ubg) ub[N * SR + o] = a.gf_raw[o] - acc;                               // raw gf (quirk q2)
    }
    if (a.write_ubg)
        for (int o = tid; o < N * NX; o += nthr) ub[(o / NX) * SR + (o % NX)] = -a.c[(size_t)b * N * NX + o];
    if (tid == 0 && a.ct_part) {
        double acc = 0.0;
        for (int j = 0; j <= N; j++) acc += a.ct_part[(size_t)b * (N + 1) + j];
        a.cost_tube[b] = sqrt(acc);
    }
}
__global__ void k_tighten(TightenArgs a) {
    const int b = blockIdx.x;
    if (a.run && !a.run[b]) return;
    tighten_dev(a, b, threadIdx.x, blockDim.x);
}

// initialize_backoff (fast_SLS_jit.py:444-454)
struct InitBackoffArgs { int B, N, NX, NU; double eps; const int *run; double *beta, *beta_f, *backoff, *backoff_f, *backoff_x, *backoff_u; int fill_beta; };
__global__ void k_init_backoff(InitBackoffArgs a) {
    const int b = blockIdx.x;
    if (a.run && !a.run[b]) return;
    const int NZ = a.NX + a.NU, NI = 2 * NZ, NIF = 2 * a.NX, N = a.N;
    const double sq = sqrt(a.eps);
    if (a.fill_beta) {   // only the handle's first solve: afterwards beta is eps wherever no sweep wrote, and k_after_qp repairs the rest
        for (int o = threadIdx.x; o < N * N * NI; o += blockDim.x) a.beta[(size_t)b * N * N * NI + o] = a.eps;
        for (int o = threadIdx.x; o < (N + 1) * NIF; o += blockDim.x) a.beta_f[(size_t)b * (N + 1) * NIF + o] = a.eps;
    }
    for (int o = threadIdx.x; o < N * NI; o += blockDim.x) a.backoff[(size_t)b * N * NI + o] = N * sq;
    for (int o = threadIdx.x; o < NIF; o += blockDim.x) a.backoff_f[(size_t)b * NIF + o] = (N + 1) * sq;
    for (int o = threadIdx.x; o < (N + 1) * a.NX; o += blockDim.x) a.backoff_x[(size_t)b * (N + 1) * a.NX + o] = 0.0;
    for (int o = threadIdx.x; o < N * a.NU; o += blockDim.x) a.backoff_u[(size_t)b * N * a.NU + o] = 0.0;
}


// ------------------------------------------------------------------------------------------------
// Batched linearisation (reference: SCP_SLS.update_jacobian, solver/SCP_SLS_jit.py:251-366):
//   A_k, B_k = d ddyn / d(x,u) at (x_k,u_k) by forward-mode AD through RK4 (one thread per (instance, stage, direction)),
//   c_k = ddyn(x_k,u_k) - x_{k+1},  g_k = g - G [x_k;u_k],  g_N = gf - Gf x_N,  q = 2 H y_nom.
// X (B,N+1,NX), U (B,N,NU) stage-major.
// ------------------------------------------------------------------------------------------------
struct LinArgs {
    int B, N;
    const double *X, *U, *g_raw, *gf_raw;
    Costs cst;
    double *A, *Bm, *c, *g, *gN, *q;
    const int *run;   // (B) 1 = linearise this instance (NULL = all)
    double *stage;    // scratch (B,N,3,NX): intermediate RK4 stage points of every (instance, stage), k_lin_val -> k_lin_tan
    double *tape;     // scratch (B,N,4,NT_MAX): the transcendental values of the four ODE evaluations of every (instance, stage) (dyn::MathRecord -> MathReplay)
};
#ifndef CLW_FN
#define CLW_FN __forceinline__      // the per-step glue of k_cl_loop (as functions called through the ABI the QP loops' allocation gets worse: 608 against 323 scratch loads)
#endif
// linearisation in two kernels: values (one thread per (instance, stage): RK4 step, its three intermediate points, the defect c_k) ...
template <int MODEL>
__device__ __forceinline__ void lin_val_item(const LinArgs &a, int b, int k) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    const size_t t = (size_t)b * a.N + k;
    const double *xg = a.X + ((size_t)b * (a.N + 1) + k) * NX, *ug = a.U + ((size_t)b * a.N + k) * NU;
    double x[NX], u[NU], st[3 * NX], f[NX];
    for (int i = 0; i < NX; i++) x[i] = xg[i];
    for (int i = 0; i < NU; i++) u[i] = ug[i];
    dyn::ddyn_stages<MODEL>(x, u, st, f, a.tape + t * 4 * dyn::NT_MAX);
    double *sg = a.stage + t * 3 * NX, *c = a.c + t * NX;
    for (int i = 0; i < 3 * NX; i++) sg[i] = st[i];
    for (int i = 0; i < NX; i++) c[i] = f[i] - xg[NX + i];
}
template <int MODEL>
__global__ __launch_bounds__(128) void k_lin_val(LinArgs a) {
    const size_t tot = (size_t)a.B * a.N;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
        const int k = t % a.N, b = t / a.N;
        if (a.run && !a.run[b]) continue;
        lin_val_item<MODEL>(a, b, k);
    }
}
// ... and tangents (one thread per (instance, stage, direction): forward-mode AD through the four ODE evaluations, stage values read back)
template <int MODEL>
__device__ __forceinline__ void lin_tan_item(const LinArgs &a, int b, int k, int dir) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    const size_t bk = (size_t)b * a.N + k;
    const double *x = a.X + ((size_t)b * (a.N + 1) + k) * NX, *u = a.U + bk * NU;
    double col[NX];
    dyn::ddyn_tangent<MODEL>(x, u, a.stage + bk * 3 * NX, a.tape + bk * 4 * dyn::NT_MAX, dir, col);
    if (dir < NX) { double *A = a.A + bk * NX * NX; for (int i = 0; i < NX; i++) A[i * NX + dir] = col[i]; }
    else { double *Bm = a.Bm + bk * NX * NU; for (int i = 0; i < NX; i++) Bm[i * NU + (dir - NX)] = col[i]; }
}
template <int MODEL>
__global__ __launch_bounds__(128) void k_lin_tan(LinArgs a) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU, NZ = NX + NU;
    const size_t tot = (size_t)a.B * a.N * NZ;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
        const int dir = t % NZ, k = (t / NZ) % a.N, b = t / ((size_t)NZ * a.N);
        if (a.run && !a.run[b]) continue;
        lin_tan_item<MODEL>(a, b, k, dir);
    }
}
template <int NX, int NU>
__device__ __forceinline__ void lin_vec_item(const LinArgs &a, int b, int e) {
    constexpr int NZ = NX + NU, NI = 2 * NZ, NIF = 2 * NX;
    const int n = NZ * a.N + NX, k = e / NZ, i = e % NZ;
    const double z = (i < NX) ? a.X[((size_t)b * (a.N + 1) + k) * NX + i] : a.U[((size_t)b * a.N + k) * NU + (i - NX)];
    if (k < a.N) {
        double *g = a.g + ((size_t)b * a.N + k) * NI;
        g[i] = a.g_raw[i] - z; g[NZ + i] = a.g_raw[NZ + i] + z;
        a.q[(size_t)b * n + e] = 2.0 * (i < NX ? a.cst.Qd[i] : a.cst.Rd[i - NX]) * z;
    } else {
        double *g = a.gN + (size_t)b * NIF;
        g[i] = a.gf_raw[i] - z; g[NX + i] = a.gf_raw[NX + i] + z;
        a.q[(size_t)b * n + e] = 2.0 * a.cst.Qfd[i] * z;
    }
}
template <int NX, int NU>
__global__ void k_lin_vec(LinArgs a) {
    constexpr int NZ = NX + NU;
    const int n = NZ * a.N + NX;
    const size_t tot = (size_t)a.B * n;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
        const int e = t % n, b = t / n;
        if (a.run && !a.run[b]) continue;
        lin_vec_item<NX, NU>(a, b, e);
    }
}
// the whole linearisation of ONE instance by one wave (the persistent closed-loop kernel k_cl_loop): same items, same arithmetic
template <int MODEL>
__device__ CLW_FN void lin_wave(const LinArgs &a, const BoundsArgs &ba, int b, int lane) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU, NZ = NX + NU;
    const int n = NZ * a.N + NX;
    if (lane < a.N) lin_val_item<MODEL>(a, b, lane);
    wla::wsync_mem();
    for (int t = lane; t < a.N * NZ; t += 64) lin_tan_item<MODEL>(a, b, t / NZ, t % NZ);
    for (int e = lane; e < n; e += 64) lin_vec_item<NX, NU>(a, b, e);
    wla::wsync_mem();
    const int mb = ba.N * (ba.NX + ba.NI) + ba.NIF;
    for (int r = lane; r < mb; r += 64) set_bounds_row(ba, b, r);
    wla::wsync_mem();
}


// ------------------------------------------------------------------------------------------------
// Closed-loop / SCP glue around the path (reference: SCP_SLS.socp_step solver/SCP_SLS_jit.py:404-473,
// reset_warm_start :500-551, expe/main_rocket_robust_closed_loop.py:149-182).  All elementwise, one thread per item.
// ------------------------------------------------------------------------------------------------
struct ClArgs {
    int B, N, NX, NU;
    double *Xn, *Un, *xmeas;          // nominal (B,N+1,NX) (B,N,NU), measured state (B,NX)
    const double *primal;             // (B,n) deviation solution of the last fast-SLS solve
    const int *success;               // (B)
    double *x0arg;                    // (B,NX) = x_nom0 - x_meas
    const double *E, *w;              // E (NX,NX) row-major (stage 0 block), w (B,NX) disturbance sample or NULL
    double *u0;                       // (B,NU) applied input
    const double *u_init;             // (NU) input used by the zero-order roll-out initialiser
};
// SCP bookkeeping of one socp_step for every instance still iterating (SCP_SLS.solve, solver/SCP_SLS_jit.py:113-135):
//   step failed                      -> the instance leaves the loop, scp_success = 0                      (:118-119)
//   step ok                          -> nominal += delta (:426-430); scp_iters = ii
//   converge mode and |delta|inf < eps -> scp_success = 1, the instance leaves the loop BEFORE the next linearisation (:123-132)
//   RTI mode                         -> scp_success = success of the last step                             (:148)
// One wave per instance; n_active counts the instances that go on.
struct ScpArgs { int ii, converge; double eps; int *active, *scp_success, *scp_iters, *n_active; double *dmax; int *updated; };
__device__ __forceinline__ void cl_scp_update_wave(const ClArgs &a, const ScpArgs &s, int b, int lane) {
    if (!s.active[b]) { if (lane == 0) s.updated[b] = 0; return; }
    const int NZ = a.NX + a.NU, n = NZ * a.N + a.NX;
    if (!a.success[b]) {
        if (lane == 0) { s.active[b] = 0; s.scp_success[b] = 0; s.scp_iters[b] = s.ii; s.updated[b] = 0; }
        return;
    }
    const double *d = a.primal + (size_t)b * n;
    double dm = 0.0;
    for (int e = lane; e < n; e += 64) {
        const int k = e / NZ, i = e % NZ;
        const double dv = d[e];
        dm = fmax(dm, fabs(dv));
        if (i < a.NX) a.Xn[((size_t)b * (a.N + 1) + k) * a.NX + i] += dv;
        else a.Un[((size_t)b * a.N + k) * a.NU + (i - a.NX)] += dv;
    }
    dm = wla::wave_max(dm);
    if (lane == 0) {
        s.dmax[b] = dm;
        s.updated[b] = 1;
        s.scp_iters[b] = s.ii;
        if (s.converge) {
            if (dm < s.eps) { s.scp_success[b] = 1; s.active[b] = 0; }
            else { s.scp_success[b] = 0; atomicAdd(s.n_active, 1); }
        } else { s.scp_success[b] = 1; atomicAdd(s.n_active, 1); }
    }
}
__global__ __launch_bounds__(64) void k_cl_scp_update(ClArgs a, ScpArgs s) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= a.B) return;
    cl_scp_update_wave(a, s, b, lane);
}
__global__ void k_cl_x0arg(ClArgs a, const int *mask = nullptr) {
    const int tot = a.B * a.NX;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += gridDim.x * blockDim.x) {
        const int b = t / a.NX, i = t % a.NX;
        if (mask && !mask[b]) continue;
        a.x0arg[t] = a.Xn[(size_t)b * (a.N + 1) * a.NX + i] - a.xmeas[t];
    }
}
// primal_infeasibility of SCP_SLS.socp_step (solver/SCP_SLS_jit.py:449-456): the signed maximum over stages and components of
// ddyn(x_k,u_k) - x_{k+1} along the nominal just updated.  One wave per instance, lane k = stage k (N <= 64).
template <int MODEL>
__device__ __forceinline__ void cl_infeas_wave(const ClArgs &a, double *pinf, int b, int k) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    const double *X = a.Xn + (size_t)b * (a.N + 1) * NX, *U = a.Un + (size_t)b * a.N * NU;
    double mx = -1e300;
    if (k < a.N) {
        double x[NX], u[NU], xp[NX];
        for (int i = 0; i < NX; i++) x[i] = X[(size_t)k * NX + i];
        for (int i = 0; i < NU; i++) u[i] = U[(size_t)k * NU + i];
        dyn::ddyn<MODEL, double>(x, u, xp);
        for (int i = 0; i < NX; i++) mx = fmax(mx, xp[i] - X[(size_t)(k + 1) * NX + i]);
    }
    mx = wla::wave_max(mx);
    if (k == 0) pinf[b] = mx;
}
template <int MODEL>
__global__ __launch_bounds__(64) void k_cl_infeas(ClArgs a, const int *updated, double *pinf) {
    const int b = blockIdx.x, k = threadIdx.x;
    if (b >= a.B || !updated[b]) return;
    cl_infeas_wave<MODEL>(a, pinf, b, k);
}
// device-side log of a closed-loop run (what the scripts store per MPC step, expe/main_rocket_robust_closed_loop.py:160-178): entry `step`
// of (B, S, ...) buffers, so a whole Monte-Carlo run needs no host round trip per step
struct ClLogArgs {
    const int *stepno, *mask;      // slsqp_cl_run: entry stepno[b] instead of `step`, only for the instances with mask[b] (NULL: all, entry `step`)
    int B, N, NX, NU, S, step;
    const double *Xn, *Un, *bx, *bu;
    const int *success, *scp_iters;
    const double *pinf;
    double *lx, *lu, *lbx, *lbu, *lstate, *lu0, *lpinf;
    int *lsucc, *lit;
};
__device__ __forceinline__ void cl_log_item(const ClLogArgs &a, int b, int o0) {
    const int nX = (a.N + 1) * a.NX, nU = a.N * a.NU;
    int o = o0;
    const size_t e = (size_t)b * a.S + (a.stepno ? min(a.stepno[b], a.S - 1) : a.step);
    if (o < nX) {
        const double v = a.Xn[(size_t)b * nX + o];
        a.lx[e * nX + o] = v;
        if (o < a.NX) a.lstate[e * a.NX + o] = v;
    } else if ((o -= nX) < nU) {
        const double v = a.Un[(size_t)b * nU + o];
        a.lu[e * nU + o] = v;
        if (o < a.NU) a.lu0[e * a.NU + o] = v;
    } else if ((o -= nU) < nX) a.lbx[e * nX + o] = a.bx[(size_t)b * nX + o];
    else { o -= nX; a.lbu[e * nU + o] = a.bu[(size_t)b * nU + o]; }
    if (o0 == 0) { a.lsucc[e] = a.success[b]; a.lit[e] = a.scp_iters[b]; a.lpinf[e] = a.pinf[b]; }
}
__global__ void k_cl_log(ClLogArgs a) {
    const int nX = (a.N + 1) * a.NX, nU = a.N * a.NU, per = 2 * nX + 2 * nU;
    const size_t tot = (size_t)a.B * per;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
        const int b = t / per;
        if (a.mask && !a.mask[b]) continue;
        cl_log_item(a, b, (int)(t % per));
    }
}
// warm-start shift (SCP_SLS_jit.py:508-518): x_k <- x_{k+1}, u_k <- u_{k+1}, u_{N-1} kept, x_N <- ddyn(x_N, u_{N-1});
// plant step (expe/main_rocket...:180-182): x_meas <- ddyn(x_meas, u0) + E w.   One thread per instance.
// slsqp_cl_run: mask (B) selects the instances; shift only those past their first step (stepno > 0); the disturbance sample of an instance is the one
// of ITS step, W_all (steps, B, NX).
template <int MODEL>
__device__ __forceinline__ void cl_plant_one(const ClArgs &a, int b, const double *w /* (B,NX) sample or NULL */) {      // one thread
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    const double *U = a.Un + (size_t)b * a.N * NU;
    double xm[NX], u[NU], xp[NX];
    for (int i = 0; i < NX; i++) xm[i] = a.xmeas[(size_t)b * NX + i];
    for (int i = 0; i < NU; i++) { u[i] = U[i]; a.u0[(size_t)b * NU + i] = u[i]; }
    dyn::ddyn<MODEL, double>(xm, u, xp);
    for (int i = 0; i < NX; i++) {
        double s = xp[i];
        if (w) for (int j = 0; j < NX; j++) s += a.E[i * NX + j] * w[(size_t)b * NX + j];
        a.xmeas[(size_t)b * NX + i] = s;
    }
}
// the new last state of the shifted nominal: ddyn(x_N, u_{N-1}) of the nominal BEFORE the shift (one thread)
template <int MODEL>
__device__ __forceinline__ void cl_shift_tail(const ClArgs &a, int b, double *xp) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    const double *X = a.Xn + (size_t)b * (a.N + 1) * NX, *U = a.Un + (size_t)b * a.N * NU;
    double xN[NX], uN[NU];
    for (int i = 0; i < NX; i++) xN[i] = X[(size_t)a.N * NX + i];
    for (int i = 0; i < NU; i++) uN[i] = U[(size_t)(a.N - 1) * NU + i];
    dyn::ddyn<MODEL, double>(xN, uN, xp);
}
template <int MODEL>
__global__ void k_cl_shift_plant(ClArgs a, int do_shift, int do_plant, const int *mask = nullptr, const int *stepno = nullptr, const double *W_all = nullptr) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    if (mask && !mask[b]) return;
    if (do_shift && stepno && stepno[b] == 0) return;
    if (W_all) a.w = W_all + (size_t)stepno[b] * a.B * NX;      // (mask[b] implies stepno[b] < steps: only instances whose chain ended in this round)
    double *X = a.Xn + (size_t)b * (a.N + 1) * NX, *U = a.Un + (size_t)b * a.N * NU;
    if (do_plant) cl_plant_one<MODEL>(a, b, a.w);
    if (do_shift) {
        double xp[NX];
        cl_shift_tail<MODEL>(a, b, xp);
        for (int k = 0; k < a.N; k++) for (int i = 0; i < NX; i++) X[(size_t)k * NX + i] = X[(size_t)(k + 1) * NX + i];
        for (int k = 0; k + 1 < a.N; k++) for (int i = 0; i < NU; i++) U[(size_t)k * NU + i] = U[(size_t)(k + 1) * NU + i];
        for (int i = 0; i < NX; i++) X[(size_t)a.N * NX + i] = xp[i];
    }
}
// the same shift of ONE instance by one wave (k_cl_loop): the tail by lane 0, the copies by all lanes (read everything, then write)
template <int MODEL>
__device__ CLW_FN void cl_shift_wave(const ClArgs &a, int b, int lane) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    double *X = a.Xn + (size_t)b * (a.N + 1) * NX, *U = a.Un + (size_t)b * a.N * NU;
    double xp[NX];
    if (lane == 0) cl_shift_tail<MODEL>(a, b, xp);
    constexpr int PX = (64 * NX + 63) / 64, PU = (63 * NU + 63) / 64;      // N <= 64: elements per lane
    double vx[PX], vu[PU];
#pragma unroll
    for (int i = 0; i < PX; i++) { const int o = lane + 64 * i; vx[i] = (o < a.N * NX) ? X[o + NX] : 0.0; }
#pragma unroll
    for (int i = 0; i < PU; i++) { const int o = lane + 64 * i; vu[i] = (o < (a.N - 1) * NU) ? U[o + NU] : 0.0; }
    wla::wsync_mem();
#pragma unroll
    for (int i = 0; i < PX; i++) { const int o = lane + 64 * i; if (o < a.N * NX) X[o] = vx[i]; }
#pragma unroll
    for (int i = 0; i < PU; i++) { const int o = lane + 64 * i; if (o < (a.N - 1) * NU) U[o] = vu[i]; }
    if (lane == 0) for (int i = 0; i < NX; i++) X[(size_t)a.N * NX + i] = xp[i];
    wla::wsync_mem();
}
// nominal initialiser replacing the reference's IPOPT call for step 0: roll-out of the plant from x_meas under a
// constant input; callers that have a better nominal pass it to slsqp_cl_init instead.
template <int MODEL>
__global__ void k_cl_rollout(ClArgs a) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    double *X = a.Xn + (size_t)b * (a.N + 1) * NX, *U = a.Un + (size_t)b * a.N * NU;
    double x[NX], u[NU], xp[NX];
    for (int i = 0; i < NX; i++) { x[i] = a.xmeas[(size_t)b * NX + i]; X[i] = x[i]; }
    for (int i = 0; i < NU; i++) u[i] = a.u_init[i];
    for (int k = 0; k < a.N; k++) {
        for (int i = 0; i < NU; i++) U[(size_t)k * NU + i] = u[i];
        dyn::ddyn<MODEL, double>(x, u, xp);
        for (int i = 0; i < NX; i++) { x[i] = xp[i]; X[(size_t)(k + 1) * NX + i] = xp[i]; }
    }
}

// ------------------------------------------------------------------------------------------------
// Nominal-trajectory initialiser.  The reference obtains the first nominal from IPOPT
// (SCP_SLS.solve_nominal_trajectory solver/SCP_SLS_jit.py:161-188; NLP of solver/nlp.py:158-217):
//     min sum_k x_k'Q x_k + u_k'R u_k + x_N'Qf x_N   s.t. x_{k+1} = ddyn(x_k,u_k), G[x_k;u_k] <= g, Gf x_N <= gf, x_0 = x_meas.
// Here the same NLP is solved per instance by sequential convex programming on the path's own QP kernel:
//   QP_j:  min (y+d)'H(y+d) + w d'd   s.t.  A dx + B du - dx+ = -tau c,   box relaxed to  hi + (1-tau) max(y-hi,0)  (same for lo)
// i.e. a step removes the fraction tau of the current constraint violation (dynamics defect c and box violation), w is a
// trust-region (proximal) weight.  Steps are judged by the ratio of actual to predicted decrease of the exact-penalty merit
// f + rho (|defect|_1 + |box violation|_1); rejected steps raise w and lower tau, good steps do the opposite.  At w -> 0,
// tau = 1 this is the plain SCP iteration of SCP_SLS without tightening, so its fixed points are KKT points of the NLP.
// State per instance S[12]: [0] w [1] kappa = 1-tau of the QP just solved / to solve [2] kappa0 (next first try) [3] f0 [4] c0 [5] v0
// [6] last ratio [7] last |d|inf.
// ------------------------------------------------------------------------------------------------
struct NomArgs {
    int B, N;
    double *Xn, *Un;
    const double *xmeas;       // (B,NX): x_0 = x_meas is a constraint of the NLP like the dynamics: |X_0 - x_meas|_1 counts as defect
    const double *primal;
    const int *qp_status;
    const double *g_raw, *gf_raw;
    Costs cst;
    double *st;
    int *active, *need_lin, *status, *iters, *n_active;
    double rho, tol, w_max;
    int mode;   // 0: merit at the current nominal (start of the iteration); 1: judge the step of the QP just solved
};
__device__ __forceinline__ double block_sum128(double v, double *red) {
    const int t = threadIdx.x;
    red[t] = v; __syncthreads();
    for (int s = 64; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    const double r = red[0]; __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max128(double v, double *red) {
    const int t = threadIdx.x;
    red[t] = v; __syncthreads();
    for (int s = 64; s > 0; s >>= 1) { if (t < s) red[t] = fmax(red[t], red[t + s]); __syncthreads(); }
    const double r = red[0]; __syncthreads();
    return r;
}
template <int MODEL>
__global__ __launch_bounds__(128) void k_nom_eval(NomArgs a) {
    constexpr int NX = dyn::Dims<MODEL>::NX, NU = dyn::Dims<MODEL>::NU, NZ = NX + NU;
    const int b = blockIdx.x, t = threadIdx.x;
    if (!a.active[b]) return;
    __shared__ double red[128];
    __shared__ int dec_s;
    const int N = a.N, n = NZ * N + NX;
    double *X = a.Xn + (size_t)b * (N + 1) * NX, *U = a.Un + (size_t)b * N * NU;
    const double *d = a.primal + (size_t)b * n;
    double *S = a.st + (size_t)b * 12;
    const bool trial = a.mode == 1;
    double f = 0.0, v = 0.0, dm = 0.0, c = 0.0;
    for (int e = t; e < n; e += 128) {
        const int k = e / NZ, i = e % NZ;
        const double z0 = (i < NX) ? X[k * NX + i] : U[k * NU + (i - NX)];
        const double dv = trial ? d[e] : 0.0, z = z0 + dv;
        const double hw = (k < N) ? (i < NX ? a.cst.Qd[i] : a.cst.Rd[i - NX]) : a.cst.Qfd[i];
        const double hi = (k < N) ? a.g_raw[i] : a.gf_raw[i], lo = (k < N) ? -a.g_raw[NZ + i] : -a.gf_raw[NX + i];
        f += hw * z * z;
        if (e >= NX) v += fmax(z - hi, 0.0) + fmax(lo - z, 0.0);   // x_0 is data (pinned to x_meas), its box is not the solver's to fix
        dm = fmax(dm, fabs(dv));
    }
    for (int k = t; k < N; k += 128) {
        double x[NX], u[NU], xp[NX];
        for (int i = 0; i < NX; i++) x[i] = X[k * NX + i] + (trial ? d[k * NZ + i] : 0.0);
        for (int i = 0; i < NU; i++) u[i] = U[k * NU + i] + (trial ? d[k * NZ + NX + i] : 0.0);
        dyn::ddyn<MODEL, double>(x, u, xp);
        for (int i = 0; i < NX; i++) c += fabs(xp[i] - (X[(k + 1) * NX + i] + (trial ? d[(k + 1) * NZ + i] : 0.0)));
    }
    if (t < NX) c += fabs(X[t] + (trial ? d[t] : 0.0) - a.xmeas[(size_t)b * NX + t]);
    f = block_sum128(f, red); v = block_sum128(v, red); c = block_sum128(c, red); dm = block_max128(dm, red);
    if (t == 0) {
        int dec = 0;   // 0 retry the QP (same linearisation), 1 step accepted, 2 converged, 3 failed
        double w = S[0], kap = S[1], kap0 = S[2];
        const double f0 = S[3], c0 = S[4], v0 = S[5];
        if (!trial) {
            S[3] = f; S[4] = c; S[5] = v; S[1] = (v > 1e-7 || c > 1e-6) ? kap0 : 0.0;
            dec = -1;
        } else {
            const int qs = a.qp_status[b];
            const double phi0 = f0 + a.rho * (c0 + v0);
            double r = 0.0;
            if (!(qs == 0 || qs == 4)) {       // QP infeasible at this tau: ask for less
                if (kap < 0.995) { kap = 1.0 - 0.3 * (1.0 - kap); dec = 0; } else dec = 3;
            } else {
                const double pred = phi0 - (f + a.rho * kap * (v0 + c0));     // linearised model: violation shrinks to kappa * (v0 + c0)
                const double act = phi0 - (f + a.rho * (c + v));
                r = pred > 0.0 ? act / pred : -1.0;
                if (pred <= 1e-12 * fmax(1.0, fabs(phi0)) || dm < a.tol) dec = (v0 < 1e-7 && c0 < 1e-7)   /* l1 sums; the QP's own 1e-10 pads on every bound add up to ~1e-9 */ ? 2 : 1;
                else if (r < 0.1) { w *= 4.0; kap = 1.0 - (1.0 - kap) / 3.0; dec = (w > a.w_max) ? 3 : 0; }
                else { dec = 1; kap0 = kap; if (r > 0.7) { w = fmax(w / 3.0, 1e-6); kap0 = kap > 0.01 ? kap / 3.0 : 0.0; } }
            }
            S[6] = r; S[7] = dm;
            if (dec == 1 || dec == 2) {
                S[3] = f; S[4] = c; S[5] = v;
                kap = (v > 1e-7 || c > 1e-6) ? kap0 : 0.0;
                a.iters[b] += 1;
            }
            S[0] = w; S[1] = kap; S[2] = kap0;
            a.need_lin[b] = (dec == 1) ? 1 : 0;
            if (dec == 2) { a.status[b] = 0; a.active[b] = 0; }
            else if (dec == 3) { a.status[b] = 2; a.active[b] = 0; }
            else atomicAdd(a.n_active, 1);
        }
        dec_s = dec;
    }
    __syncthreads();
    if (dec_s == 1 || dec_s == 2) {
        for (int e = t; e < n; e += 128) {
            const int k = e / NZ, i = e % NZ;
            if (i < NX) X[k * NX + i] += d[e]; else U[k * NU + (i - NX)] += d[e];
        }
    }
}
// bounds of the initialiser's QP: dynamics rows -tau c (+-eps), box rows g_k + kappa max(-g_k, 0) (g_k = g - G y, so max(-g_k,0) is the
// current violation); the box of x_0 is dropped (x_0 is pinned).
struct NomBoundsArgs { int B, N, NX, NI, NIF; const double *g, *gN, *c, *st; const int *run; double *ubg, *lbg; double eps; };
__global__ void k_nom_bounds(NomBoundsArgs a) {
    const int SR = a.NX + a.NI, mb = a.N * SR + a.NIF, NZ = a.NI / 2;
    const size_t tot = (size_t)a.B * mb;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = idx / mb, r = idx % mb;
        if (a.run && !a.run[b]) continue;
        const double kap = a.st[(size_t)b * 12 + 1];
        double u, l;
        if (r < a.N * SR) {
            const int k = r / SR, i = r % SR;
            if (i < a.NX) { const double cv = (1.0 - kap) * a.c[((size_t)b * a.N + k) * a.NX + i]; u = -cv + a.eps; l = -cv - a.eps; }
            else {
                const int j = i - a.NX;
                const double gv = a.g[((size_t)b * a.N + k) * a.NI + j];
                u = (k == 0 && (j % NZ) < a.NX) ? 1e20 : gv + kap * fmax(-gv, 0.0) + a.eps; l = -1e20;
            }
        } else { const double gv = a.gN[(size_t)b * a.NIF + (r - a.N * SR)]; u = gv + kap * fmax(-gv, 0.0) + a.eps; l = -1e20; }
        a.ubg[idx] = u; a.lbg[idx] = l;
    }
}
// pin of the initialiser's QP: d_x0 = tau (x_meas - X_0), the same fraction of this constraint's violation as of the others
__global__ void k_nom_x0(int B, int N, int NX, const double *Xn, const double *xmeas, const double *st, double *x0val) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * NX) return;
    const int b = t / NX, i = t % NX;
    x0val[t] = (1.0 - st[(size_t)b * 12 + 1]) * (xmeas[t] - Xn[(size_t)b * (N + 1) * NX + i]);
}
__global__ void k_nom_init(int B, double *st, int *active, int *need_lin, int *status, int *iters, double w0, double kappa0) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double *S = st + (size_t)b * 12;
    for (int i = 0; i < 12; i++) S[i] = 0.0;
    S[0] = w0; S[1] = kappa0; S[2] = kappa0;
    active[b] = 1; need_lin[b] = 1; status[b] = 1; iters[b] = 0;
}

}  // namespace slsqp
