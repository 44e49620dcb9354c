// wave_la.hpp -- single-wavefront (64 lanes) dense linear algebra on LDS-resident small blocks, fp64.
//
// Every routine is executed by ONE wave (the kernels in this package launch 64-thread workgroups, one
// MPC instance or one (instance, disturbance column) per wave), operands live in LDS, sizes are
// compile-time so all loops unroll and LDS addresses fold to immediates.  `wsync()` orders LDS traffic
// between lanes of the wave (s_waitcnt lgkmcnt(0) + s_barrier; with a one-wave workgroup the barrier
// itself is free).
#pragma once
#include <hip/hip_runtime.h>

namespace wla {

#ifndef NE_KUNROLL
#define NE_KUNROLL 32
#endif
template <typename T> struct ident { using type = T; };
template <typename T> using ident_t = typename ident<T>::type;

// LDS-only ordering between lanes of the (single) wave of a workgroup.  NOT __syncthreads(): its workgroup fence
// also waits for every outstanding global load/store (vmcnt(0)), which cost ~86 % of the wave's lifetime in the first
// profile.  LDS instructions of one wave execute in order, so waiting for lgkmcnt(0) and stopping the compiler from
// moving memory operations across this point is all that is needed.
// The DS instructions of a wave are executed by the LDS in issue order, so a later read of another lane's earlier write needs no
// s_waitcnt (the compiler still inserts the waits that register results need): wsync() only has to stop the COMPILER from moving
// LDS accesses across it.  -DWLA_WSYNC_WAITCNT restores the explicit drain (measured: 2 % slower, same bits).
#ifdef WLA_WSYNC_WAITCNT
__device__ __forceinline__ void wsync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void wsync() { asm volatile("" ::: "memory"); }
#endif
// Full version for hand-offs through global memory between lanes of the wave (phase boundaries only).
__device__ __forceinline__ void wsync_mem() { __threadfence_block(); __syncthreads(); }

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_or(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}

// Two consecutive doubles as one 16-byte move.  The addresses are only 8-byte aligned (stage blocks of odd size follow each other): from global
// memory this is one global_load_dwordx4 / global_store_dwordx4 (dword alignment is all the hardware asks for), to LDS one ds_write2_b64.
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 ld2(const double *p) { d2 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st2(double *p, d2 v) { __builtin_memcpy(p, &v, 16); }

// C(MxN) = alpha * A(MxK) * B(NxK)' with RBxCB register blocking per lane (one pass, needs ceil(M/RB)*ceil(N/CB) <= 64).
// Out-of-range rows/cols are clamped on load and masked on store.
template <int M, int N, int K, int RB, int CB, typename T>
__device__ __forceinline__ void gemm_nt_blk(const T *A, int lda, const T *B, int ldb, T *C, int ldc, ident_t<T> alpha, int lane) {
    constexpr int TR = (M + RB - 1) / RB, TC = (N + CB - 1) / CB;
    static_assert(TR * TC <= 64, "one pass only");
    if (lane < TR * TC) {
        const int tr = lane / TC, tc = lane % TC;
        int ri[RB], cj[CB];
#pragma unroll
        for (int r = 0; r < RB; r++) ri[r] = min(tr * RB + r, M - 1);
#pragma unroll
        for (int q = 0; q < CB; q++) cj[q] = min(tc * CB + q, N - 1);
        T acc[RB][CB];
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int q = 0; q < CB; q++) acc[r][q] = T(0);
#pragma unroll NE_KUNROLL
        for (int k = 0; k < K; k++) {
            T a[RB], b[CB];
#pragma unroll
            for (int r = 0; r < RB; r++) a[r] = A[ri[r] * lda + k];
#pragma unroll
            for (int q = 0; q < CB; q++) b[q] = B[cj[q] * ldb + k];
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int q = 0; q < CB; q++) acc[r][q] = fma(a[r], b[q], acc[r][q]);
        }
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int q = 0; q < CB; q++)
                if (tr * RB + r < M && tc * CB + q < N) C[(tr * RB + r) * ldc + tc * CB + q] = alpha * acc[r][q];
    }
}


#ifndef GEMM_BLK_KUNROLL
#define GEMM_BLK_KUNROLL 1
#endif
// General register-blocked product C(MxN) = alpha * op(A) op(B) [+ C], RBxCB outputs per lane, as many passes as needed.
template <int M, int N, int K, bool TA, bool TB, int RB, int CB, bool ACC>
__device__ __forceinline__ void gemm_blk(const double *A, int lda, const double *B, int ldb, double *C, int ldc, double alpha, int lane) {
    constexpr int TR = (M + RB - 1) / RB, TC = (N + CB - 1) / CB, NTASK = TR * TC;
#pragma unroll
    for (int t0 = 0; t0 < NTASK; t0 += 64) {
        const int t = t0 + lane;
        if (t < NTASK) {
            const int tr = t / TC, tc = t % TC;
            int ri[RB], cj[CB];
#pragma unroll
            for (int r = 0; r < RB; r++) ri[r] = min(tr * RB + r, M - 1);
#pragma unroll
            for (int q = 0; q < CB; q++) cj[q] = min(tc * CB + q, N - 1);
            double acc[RB][CB];
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int q = 0; q < CB; q++) acc[r][q] = 0.0;
            // k loop deliberately NOT unrolled: fully unrolled, the compiler hoists all K x (RB + CB) LDS loads (244 VGPRs in k_sweep,
            // 2 waves/SIMD); rolled it needs 118 and the sweep runs 3 waves/SIMD, 17 % faster (measured)
#pragma unroll GEMM_BLK_KUNROLL
            for (int k = 0; k < K; k++) {
                double a[RB], b[CB];
#pragma unroll
                for (int r = 0; r < RB; r++) a[r] = TA ? A[k * lda + ri[r]] : A[ri[r] * lda + k];
#pragma unroll
                for (int q = 0; q < CB; q++) b[q] = TB ? B[cj[q] * ldb + k] : B[k * ldb + cj[q]];
#pragma unroll
                for (int r = 0; r < RB; r++)
#pragma unroll
                    for (int q = 0; q < CB; q++) acc[r][q] = fma(a[r], b[q], acc[r][q]);
            }
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int q = 0; q < CB; q++)
                    if (tr * RB + r < M && tc * CB + q < N) {
                        double *c = C + (tr * RB + r) * ldc + tc * CB + q;
                        *c = ACC ? fma(alpha, acc[r][q], *c) : alpha * acc[r][q];
                    }
        }
    }
}

// C(MxN) = op(A)(MxK) op(B)(KxN) for M, N, K <= 17 on the fp64 matrix core: the 16x16x16 core of the product is four
// v_mfma_f64_16x16x4_f64 (lane l feeds A[l&15][k0 + (l>>4)] and B[k0 + (l>>4)][l&15]; its 4 results are rows (l>>4) + 4r, column l&15), the
// 17th k adds a rank-1 term to those results, and the 17th row / column of the result (33 entries) are plain dot products on the
// vector ALU, which runs beside the matrix pipe.  Needs all 64 lanes active.  Summation order differs from gemm_blk (last-bit effects).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
// D (optional, MxN, leading dim ldd): C = D + op(A) op(B).
// sk (optional, K): op(A) is used as op(A) diag(sk).
template <int M, int N, int K, bool TA, bool TB, bool ADD = false, bool SCALEK = false>
__device__ __forceinline__ void gemm_mfma(const double *A, int lda, const double *B, int ldb, double *C, int ldc, int lane, const double *D = nullptr,
                                          int ldd = 0, const double *sk = nullptr) {
    static_assert(M <= 17 && N <= 17 && K <= 17, "one 16x16 tile plus one border row / column / k");
    constexpr int MC = M < 16 ? M : 16, NC = N < 16 ? N : 16, KC = K < 16 ? K : 16;
    asm volatile("" : "+v"(lane));      // indices derived from the lane id are recomputed per call, not hoisted out of the caller's loops and kept live (see spd_inv_gj_mfma)
    const int li = lane & 15, lk = lane >> 4;
    auto a_at = [&](int i, int k) -> double { const double v = TA ? A[k * lda + i] : A[i * lda + k]; return SCALEK ? v * sk[k] : v; };
    auto b_at = [&](int k, int j) -> double { return TB ? B[j * ldb + k] : B[k * ldb + j]; };
    const int ia = min(li, MC - 1), jb = min(li, NC - 1);
    mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
    if constexpr (ADD) {
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = D[min(lk + 4 * r, MC - 1) * ldd + jb];
    }
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += 4) {
        const int k = k0 + lk, kc = min(k, KC - 1);
        const double av = a_at(ia, kc), bv = b_at(kc, jb);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((li < MC && k < KC) ? av : 0.0, (li < NC && k < KC) ? bv : 0.0, acc, 0, 0, 0);
    }
    if constexpr (K == 17) {
        const double b16 = b_at(16, jb);
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = fma(a_at(min(lk + 4 * r, MC - 1), 16), b16, acc[r]);
    }
    // every load (core, rank-1, border) is issued before the first store, so C may be A or B itself: the LDS operations of the one wave
    // of the workgroup execute in program order.
    // Border entries: row 16 of the result (N entries, with the corner) and column 16 (MC entries) are vector-matrix / matrix-vector products;
    // each is split three ways over k across lane groups (lane = entry + count * g) and combined through the crossbar: 6 dependent FMAs and
    // 12 LDS reads per lane instead of the 17 / 34 of one lane per entry.
    double srow = 0.0, scol = 0.0;
    if constexpr (M == 17) {
        const int g = lane / N, j = lane - g * N;
        double v = 0.0;
        if (g < 3) {
            if (ADD && g == 0) v = D[16 * ldd + j];
#pragma unroll
            for (int q = 0; q < (K + 2) / 3; q++) { const int k = 3 * q + g; if (k < K) v = fma(a_at(16, k), b_at(k, j), v); }
        }
        srow = v + __shfl(v, lane + N) + __shfl(v, lane + 2 * N);
    }
    if constexpr (N == 17) {
        const int g = lane / MC, i = lane - g * MC;
        double v = 0.0;
        if (g < 3) {
            if (ADD && g == 0) v = D[i * ldd + 16];
#pragma unroll
            for (int q = 0; q < (K + 2) / 3; q++) { const int k = 3 * q + g; if (k < K) v = fma(a_at(i, k), b_at(k, 16), v); }
        }
        scol = v + __shfl(v, lane + MC) + __shfl(v, lane + 2 * MC);
    }
    if constexpr (M == 17 || N == 17) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; r++) { const int i = lk + 4 * r; if (i < MC && li < NC) C[i * ldc + li] = acc[r]; }
    if constexpr (M == 17) { if (lane < N) C[16 * ldc + lane] = srow; }
    if constexpr (N == 17) { if (lane < MC) C[lane * ldc + 16] = scol; }
}

// Two products with one right-hand operand, for the SLS propagation step:  Cu (MU x N) = Ku (MU x K) P  and  Cx (MX x N) = Ax (MX x K) P,
// P (K x N) shared.  Same arithmetic as two gemm_mfma calls (core: four MFMAs each, 17th k as a rank-1 term, 17th row / column as three-way
// split vector products), but the two accumulation chains are independent and interleave on the matrix pipe, and P's operand is read once.
// MU <= 16; MX, N, K <= 17; no transposes; all 64 lanes active; Cu, Cx must not alias P.
template <int MU, int MX, int N, int K>
__device__ __forceinline__ void gemm_mfma_pair(const double *Ku, int ldk, const double *Ax, int lda, const double *P, int ldp, double *Cu, int ldcu, double *Cx,
                                               int ldcx, int lane) {
    static_assert(MU <= 16 && MX <= 17 && N <= 17 && K <= 17, "one 16x16 tile plus one border row / column / k");
    constexpr int MUC = MU, MXC = MX < 16 ? MX : 16, NC = N < 16 ? N : 16, KC = K < 16 ? K : 16;
    const int li = lane & 15, lk = lane >> 4;
    const int iu = min(li, MUC - 1), ix = min(li, MXC - 1), jb = min(li, NC - 1);
    mfma_d4 au = {0.0, 0.0, 0.0, 0.0}, ax = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += 4) {
        const int k = k0 + lk, kc = min(k, KC - 1);
        const bool kok = k < KC;
        const double bv = (li < NC && kok) ? P[kc * ldp + jb] : 0.0;
        au = __builtin_amdgcn_mfma_f64_16x16x4f64((li < MUC && kok) ? Ku[iu * ldk + kc] : 0.0, bv, au, 0, 0, 0);
        ax = __builtin_amdgcn_mfma_f64_16x16x4f64((li < MXC && kok) ? Ax[ix * lda + kc] : 0.0, bv, ax, 0, 0, 0);
    }
    if constexpr (K == 17) {
        const double b16 = P[16 * ldp + jb];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            au[r] = fma(Ku[min(lk + 4 * r, MUC - 1) * ldk + 16], b16, au[r]);
            ax[r] = fma(Ax[min(lk + 4 * r, MXC - 1) * lda + 16], b16, ax[r]);
        }
    }
    double srow = 0.0, scx = 0.0, scu = 0.0;
    if constexpr (MX == 17) {       // row 16 of Cx (with the corner)
        const int g = lane / N, j = lane - g * N;
        double v = 0.0;
        if (g < 3) {
#pragma unroll
            for (int q = 0; q < (K + 2) / 3; q++) { const int k = 3 * q + g; if (k < K) v = fma(Ax[16 * lda + k], P[k * ldp + j], v); }
        }
        srow = v + __shfl(v, lane + N) + __shfl(v, lane + 2 * N);
    }
    if constexpr (N == 17) {        // column 16 of Cx (rows < 16) and of Cu, in one pass: entries MXC + MUC <= 21, three lane groups
        constexpr int NE = MXC + MUC;
        static_assert(3 * NE <= 64, "three lane groups");
        const int g = lane / NE, e = lane - g * NE;
        const double *row = (e < MXC) ? Ax + e * lda : Ku + (e - MXC) * ldk;
        double v = 0.0;
        if (g < 3) {
#pragma unroll
            for (int q = 0; q < (K + 2) / 3; q++) { const int k = 3 * q + g; if (k < K) v = fma(row[k], P[k * ldp + 16], v); }
        }
        const double t = v + __shfl(v, lane + NE) + __shfl(v, lane + 2 * NE);
        scx = t; scu = t;
    }
    if constexpr (MX == 17 || N == 17) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = lk + 4 * r;
        if (i < MUC && li < NC) Cu[i * ldcu + li] = au[r];
        if (i < MXC && li < NC) Cx[i * ldcx + li] = ax[r];
    }
    if constexpr (MX == 17) { if (lane < N) Cx[16 * ldcx + lane] = srow; }
    if constexpr (N == 17) {
        if (lane < MXC) Cx[lane * ldcx + 16] = scx;
        else if (lane < MXC + MUC) Cu[(lane - MXC) * ldcu + 16] = scu;
    }
}

// Lower triangle (i >= j) of  Y = M1 A' + B diag(piu) B' - T M1' + diag(d) ,  all NX x NX (B: NX x NU), 2x2 blocks, one pass.
// useT = false drops the T term.  Only the lower triangle of Y is written (the Cholesky reads nothing else).
template <int NX, int NU, typename R>
__device__ __forceinline__ void build_Y_lower(const R *M1, const R *A, const R *B, const R *piu, const R *Tm,
                                              bool useT, const R *d, ident_t<R> delta, R *Y, int lane) {
    constexpr int T = (NX + 1) / 2, NT = T * (T + 1) / 2;
    static_assert(NT <= 64, "one pass only");
    if (lane < NT) {
        // lane -> (bi >= bj) in the lower-triangular block grid
        int bi = 0, rem = lane;
        while (rem > bi) { rem -= bi + 1; bi++; }
        const int bj = rem;
        const int i0 = bi * 2, i1 = min(i0 + 1, NX - 1), j0 = bj * 2, j1 = min(j0 + 1, NX - 1);
        R a00 = 0, a01 = 0, a10 = 0, a11 = 0;
#pragma unroll NE_KUNROLL
        for (int k = 0; k < NX; k++) {
            const R x0 = M1[i0 * NX + k], x1 = M1[i1 * NX + k], y0 = A[j0 * NX + k], y1 = A[j1 * NX + k];
            a00 = fma(x0, y0, a00); a01 = fma(x0, y1, a01); a10 = fma(x1, y0, a10); a11 = fma(x1, y1, a11);
        }
#pragma unroll
        for (int k = 0; k < NU; k++) {
            const R pk = piu[k];
            const R x0 = B[i0 * NU + k] * pk, x1 = B[i1 * NU + k] * pk, y0 = B[j0 * NU + k], y1 = B[j1 * NU + k];
            a00 = fma(x0, y0, a00); a01 = fma(x0, y1, a01); a10 = fma(x1, y0, a10); a11 = fma(x1, y1, a11);
        }
        if (useT) {   // - T M1'  (T = M1 Dinv_prev; the product is symmetric)
#pragma unroll NE_KUNROLL
            for (int k = 0; k < NX; k++) {
                const R x0 = Tm[i0 * NX + k], x1 = Tm[i1 * NX + k], y0 = M1[j0 * NX + k], y1 = M1[j1 * NX + k];
                a00 = fma(-x0, y0, a00); a01 = fma(-x0, y1, a01); a10 = fma(-x1, y0, a10); a11 = fma(-x1, y1, a11);
            }
        }
        if (bi == bj) { a00 += d[i0] + delta; if (i0 + 1 < NX) a11 += d[i0 + 1] + delta; }
        Y[i0 * NX + j0] = a00;
        if (j0 + 1 < NX && i0 >= j0 + 1) Y[i0 * NX + j0 + 1] = a01;
        if (i0 + 1 < NX) {
            Y[(i0 + 1) * NX + j0] = a10;
            if (j0 + 1 < NX) Y[(i0 + 1) * NX + j0 + 1] = a11;
        }
    }
}

// build_Y_lower on the fp64 matrix core (NX in 13..17, NU <= 4) with M1 = A diag(pix) formed on the fly: the three products accumulate in
// one 16x16 tile (9 MFMAs), the 17th k is a rank-1 term on the results, and row 16 of the result (the only border entries in the lower
// triangle) is three partial dot products per entry on 51 lanes, summed through the crossbar.  useM = false drops the M1 terms (stage 0).
// Writes the lower triangle of the core and the whole last row; every load precedes the first store, so Y may be Tm's buffer.
template <int NX, int NU>
__device__ __forceinline__ void build_Y_mfma(const double *A, const double *pix, const double *B, const double *piu, const double *Tm, bool useM,
                                             const double *d, double delta, double *Y, int lane) {
    static_assert(NX >= 5 && NX <= 17 && NU <= 4, "one 16x16 tile");
    constexpr int MC = NX < 16 ? NX : 16;
    asm volatile("" : "+v"(lane));      // indices derived from the lane id are recomputed per call, not hoisted out of the caller's loops and kept live (see spd_inv_gj_mfma)
    const int li = lane & 15, lk = lane >> 4, lc = min(li, MC - 1);
    mfma_d4 acc;
#pragma unroll
    for (int r = 0; r < 4; r++) { const int i = lk + 4 * r; acc[r] = (i == li && i < MC) ? d[min(i, NX - 1)] + delta : 0.0; }
    if (useM) {
#pragma unroll
        for (int k0 = 0; k0 < MC; k0 += 4) {
            const int k = k0 + lk, kc = min(k, MC - 1);
            const bool ok = (li < MC) && (k < MC);
            const double av = A[lc * NX + kc], m1 = av * pix[kc], tv = Tm[lc * NX + kc];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? m1 : 0.0, ok ? av : 0.0, acc, 0, 0, 0);              // M1 A'
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? -tv : 0.0, ok ? m1 : 0.0, acc, 0, 0, 0);            // - T M1'
        }
    }
    {
        const int u = min(lk, NU - 1);
        const bool ok = (li < MC) && (lk < NU);
        const double bv = B[lc * NU + u];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? bv * piu[u] : 0.0, ok ? bv : 0.0, acc, 0, 0, 0);     // B diag(piu) B'
    }
    if constexpr (NX == 17) {
        if (useM) {
            const double a16 = A[lc * NX + 16], p16 = pix[16], m16 = a16 * p16;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = lk + 4 * r;
                acc[r] = fma(A[i * NX + 16] * p16, a16, acc[r]);
                acc[r] = fma(-Tm[i * NX + 16], m16, acc[r]);
            }
        }
    }
    double srow = 0.0;      // row 16 of the result, computed before anything is stored
    if constexpr (NX == 17) {
        const int g = lane / 17, j = lane % 17;
        double v = 0.0;
        if (g == 0) {
            if (useM) {
#pragma unroll
                for (int k = 0; k < NX; k++) v = fma(A[16 * NX + k] * pix[k], A[j * NX + k], v);
            }
        } else if (g == 1) {
            if (useM) {
#pragma unroll
                for (int k = 0; k < NX; k++) v = fma(-Tm[16 * NX + k], A[j * NX + k] * pix[k], v);
            }
        } else if (g == 2) {
#pragma unroll
            for (int u = 0; u < NU; u++) v = fma(B[16 * NU + u] * piu[u], B[j * NU + u], v);
            if (j == 16) v += d[16] + delta;
        }
        srow = v + __shfl(v, lane + 17) + __shfl(v, lane + 34);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { const int i = lk + 4 * r; if (i < MC && li <= i) Y[i * NX + li] = acc[r]; }
    if constexpr (NX == 17) {
        if (lane < 17) Y[16 * NX + lane] = srow;
    }
}

// y(M) = op(A)(MxK) x(K) with the sum over k split three ways across lane groups (lane = i + M g, g = 0..2) and combined through the
// crossbar: a dependent chain of ceil(K/3) FMAs instead of K, and a third of the LDS reads per lane.  Valid in lanes < M.
template <int M, int K, bool TA, typename T>
__device__ __forceinline__ T matvec_split3(const T *A, int lda, const T *x, int lane) {
    static_assert(3 * M <= 64, "three lane groups");
    const int g = lane / M, i = lane - g * M;
    T s = T(0);
    if (g < 3) {
#pragma unroll
        for (int q = 0; q < (K + 2) / 3; q++) {
            const int k = 3 * q + g;
            if (k < K) s = fma(TA ? A[k * lda + i] : A[i * lda + k], x[k], s);
        }
    }
    return s + __shfl(s, lane + M) + __shfl(s, lane + 2 * M);
}

// v_rcp_f64 / v_rsq_f64 seeds + two Newton steps: ~1 ulp, a dependent chain of ~8 instructions instead of the ~30 of the
// correctly rounded division / sqrt sequences (the factorisation below is latency-bound on exactly these chains).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsq(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    r = r * fma(-h * r, r, 1.5);
    r = r * fma(-h * r, r, 1.5);
    return r;
}
__device__ __forceinline__ float fast_rcp(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float readlane_d(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float bperm_d(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }
template <typename T> __device__ __forceinline__ T tiny_pivot();
template <> __device__ __forceinline__ double tiny_pivot<double>() { return 1e-300; }
template <> __device__ __forceinline__ float tiny_pivot<float>() { return 1e-30f; }
// broadcast lane `l` (compile-time / wave-uniform) of a double through SGPRs: no LDS, no waitcnt
__device__ __forceinline__ double readlane_d(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}


// Inverse of the MxM SPD matrix Y (lower triangle in LDS) by the symmetric Gauss-Jordan sweep, distributed 2-D over the wave:
// lane (bi >= bj) owns the 2x2 block rows {2bi,2bi+1} x cols {2bj,2bj+1} of the lower triangle in registers; per pivot j the
// pivot column is exchanged through `col` (LDS, M+2 doubles) and every lane does 4 FMAs.  17 short steps on 45 lanes instead
// of 17 long ones on 17 lanes (the row-per-lane Cholesky + triangular inverse this replaces).  Dinv: full symmetric MxM in LDS.
// Returns non-zero (wave-uniform) if a pivot was not positive (pivot clamped).
// arbitrary-lane gather of a double through the LDS crossbar (ds_bpermute_b32 x2): no LDS memory, no write+wait round trip
__device__ __forceinline__ double bperm_d(double v, int src_lane) {
    const int a = src_lane << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// lane (bi >= bj) of the lower-triangular 2x2 block grid -> its block of a symmetric MxM matrix, written mirrored to the full matrix in LDS
template <int M> constexpr int gj_blocks() { return ((M + 1) / 2) * ((M + 1) / 2 + 1) / 2; }
// row bi of lane l = bi (bi + 1) / 2 + bj, bj <= bi, in the lower-triangular block grid.  Deliberately a loop: a closed form makes the block
// indices loop-invariant for the compiler, which then hoists every address derived from them out of the stage loop of the sweeps and keeps them in
// registers across the factorisation (k_qp_solve: 57 -> 194 spilled VGPRs, the kernel twice as slow -- measured).
__device__ __forceinline__ int tri_row(int l) {
    int bi = 0, rem = l;
    while (rem > bi) { rem -= bi + 1; bi++; }
    return bi;
}
template <int M, typename R>
__device__ __forceinline__ void gj_blocks_to_lds(R v00, R v01, R v10, R v11, R *Dinv, int ldi, int lane) {
    constexpr int NT = gj_blocks<M>();
    if (lane < NT) {
        const int bi = tri_row(lane), bj = lane - bi * (bi + 1) / 2, i0 = 2 * bi, i1 = i0 + 1, l0 = 2 * bj, l1 = l0 + 1;
        if (i0 < M && l0 < M) { Dinv[i0 * ldi + l0] = v00; Dinv[l0 * ldi + i0] = v00; }
        if (i0 < M && l1 < M && bi != bj) { Dinv[i0 * ldi + l1] = v01; Dinv[l1 * ldi + i0] = v01; }
        if (i1 < M && l0 < M) { Dinv[i1 * ldi + l0] = v10; Dinv[l0 * ldi + i1] = v10; }
        if (i1 < M && l1 < M) { Dinv[i1 * ldi + l1] = v11; Dinv[l1 * ldi + i1] = v11; }
    }
}

template <int M, typename R>
__device__ __forceinline__ int spd_inv_gj(const R *Y, int ld, R *Dinv, int ldi, R *blk, int lane) {
    // Block (2x2 pivots) symmetric Gauss-Jordan sweep: T = ceil(M/2) rounds instead of M; branch-free.
    //   P = A_JJ ;  A_IL -= A_IJ P^-1 A_JL (I,L not J) ;  A_IJ <- A_IJ P^-1 ;  A_JL <- P^-1 A_JL ;  A_JJ <- -P^-1 ;  result = -A^-1
    constexpr int T = (M + 1) / 2, NT = T * (T + 1) / 2;
    static_assert(NT <= 64, "one wave");
    const int lt = min(lane, NT - 1), bi = tri_row(lt), bj = lt - bi * (bi + 1) / 2;
    const bool act = lane < NT;
    const int i0 = 2 * bi, i1 = i0 + 1, l0 = 2 * bj, l1 = l0 + 1;
    // padded (2T x 2T) matrix: identity in the padding row/col when M is odd
    auto ld_el = [&](int i, int l) -> R {
        const bool pad = (i >= M) | (l >= M);
        const int ii = min(i, M - 1), ll = min(l, M - 1);
        const R v = (ii >= ll) ? Y[ii * ld + ll] : Y[ll * ld + ii];
        return pad ? ((i == l) ? R(1) : R(0)) : v;
    };
    R a00 = ld_el(i0, l0), a01 = ld_el(i0, l1), a10 = ld_el(i1, l0), a11 = ld_el(i1, l1);
    const int tri_bi = bi * (bi + 1) / 2, tri_bj = bj * (bj + 1) / 2;
    int fail = 0;
#pragma unroll
    for (int jb = 0; jb < T; jb++) {
        const int tri_jb = jb * (jb + 1) / 2, dl = tri_jb + jb;
        // One uniform update for every lane:  A <- m A - (C_I P^-1) C_L'  with C_I = A_{I,J}, C_L = A_{L,J} fetched through the crossbar.
        // The diagonal lane (the only one pivot-row lanes read as C_I and pivot-column lanes read as C_L) sends -I instead of its block, and the
        // lanes of the pivot row / column start from m = 0:  column: 0 - (A_IJ P^-1)(-I) = A_IJ P^-1;  row: 0 - (-P^-1) A_JL = P^-1 A_JL;
        // diagonal: 0 - (-P^-1)(-I) = -P^-1.  (The pivot block itself travels through v_readlane below.)
        const bool isD = (bi == jb) & (bj == jb), inRC = (bi == jb) | (bj == jb);
        const R md = isD ? R(0) : R(1), cd = isD ? R(-1) : R(0), m = inRC ? R(0) : R(1);
        const R s00 = fma(a00, md, cd), s01 = a01 * md, s10 = a10 * md, s11 = fma(a11, md, cd);
        // Issued first so the crossbar round trip overlaps the dependent reciprocal chain of the pivot block below.
        const int srcI = (bi >= jb) ? tri_bi + jb : tri_jb + bi, srcL = (bj >= jb) ? tri_bj + jb : tri_jb + bj;
        const R f00 = bperm_d(s00, srcI), f01 = bperm_d(s01, srcI), f10 = bperm_d(s10, srcI), f11 = bperm_d(s11, srcI);
        const R g00 = bperm_d(s00, srcL), g01 = bperm_d(s01, srcL), g10 = bperm_d(s10, srcL), g11 = bperm_d(s11, srcL);
        __builtin_amdgcn_sched_barrier(0);
        // pivot block (uniform): P = [[pa, pb],[pb, pc]] from the diagonal lane; P^-1 by two scalar eliminations (as stable as 1x1 pivots)
        const R pa = readlane_d(a00, dl), pb = readlane_d(a10, dl), pc = readlane_d(a11, dl);
        R d1 = pa;
        if (!(d1 > tiny_pivot<R>())) { fail = 1; d1 = tiny_pivot<R>(); }
        const R r1 = fast_rcp(d1), bp = pb * r1;
        R d2 = fma(-pb, bp, pc);
        if (!(d2 > tiny_pivot<R>())) { fail = 1; d2 = tiny_pivot<R>(); }
        const R r2 = fast_rcp(d2);
        const R q11 = r2, q01 = -bp * r2, q00 = fma(bp * bp, r2, r1);      // P^-1 = [[q00, q01],[q01, q11]]
        const bool tI = bi < jb, tL = bj < jb;
        const R ci00 = f00, ci01 = tI ? f10 : f01, ci10 = tI ? f01 : f10, ci11 = f11;
        const R cl00 = g00, cl01 = tL ? g10 : g01, cl10 = tL ? g01 : g10, cl11 = g11;
        const R t00 = fma(ci00, q00, ci01 * q01), t01 = fma(ci00, q01, ci01 * q11);
        const R t10 = fma(ci10, q00, ci11 * q01), t11 = fma(ci10, q01, ci11 * q11);
        a00 = fma(a00, m, -fma(t00, cl00, t01 * cl01)); a01 = fma(a01, m, -fma(t00, cl10, t01 * cl11));
        a10 = fma(a10, m, -fma(t10, cl00, t11 * cl01)); a11 = fma(a11, m, -fma(t10, cl10, t11 * cl11));
    }
    // Dinv = -swept: mirrored to the full matrix in LDS, and (blk != nullptr) the lane's 2x2 block as it is to the block-packed copy in global
    // memory -- 4 NT doubles per matrix (180 of 289 for M = 17), two 16-byte stores per lane straight from the registers
    gj_blocks_to_lds<M>(-a00, -a01, -a10, -a11, Dinv, ldi, lane);
    if (blk != nullptr && act) {
        if constexpr (sizeof(R) == 8) { st2((double *)blk + 4 * lane, d2{-a00, -a01}); st2((double *)blk + 4 * lane + 2, d2{-a10, -a11}); }
        else { blk[4 * lane] = -a00; blk[4 * lane + 1] = -a01; blk[4 * lane + 2] = -a10; blk[4 * lane + 3] = -a11; }
    }
    wsync();
    return fail;
}

// The same symmetric Gauss-Jordan sweep with 2x2 pivots for 13 <= M <= 17 on the fp64 matrix core.  The 16x16 core of the matrix lives in the
// accumulator layout of v_mfma_f64_16x16x4_f64 (lane l: column l & 15, rows (l >> 4) + 4 r), row / column 16 of a 17 x 17 matrix in one value per
// lane plus the corner.  One round = one rank-2 update of the whole tile:  A <- m o A - (C P^-1) C'  with C the two pivot columns -- which, the
// matrix being symmetric, are the two pivot ROWS: lane (col, .) fetches rows 2jb, 2jb+1 at its column from the accumulators of two other lanes
// (two crossbar gathers; the 2x2-block version needs eight), forms its entries of both operands from them, zeroes its accumulators in the pivot
// rows / columns (the -I substitution of spd_inv_gj makes the one update produce C P^-1, P^-1 C' and -P^-1 there) and issues ONE MFMA.  The border
// row follows with two fused multiply-adds per lane, the last round pivots on the corner.  ~620 instead of ~1 090 instructions for M = 17, same
// pivots, same formulas; sums inside a rank-2 update are associated differently (last-bit differences).  Needs all 64 lanes.
// Y: lower triangle in LDS (ld); Dinv: full symmetric M x M in LDS (ldi); blk (optional): block-packed copy in global memory as spd_inv_gj writes it.
template <int M>
__device__ __forceinline__ int spd_inv_gj_mfma(const double *Y, int ld, double *Dinv, int ldi, double *blk, int lane) {
    static_assert(M >= 9 && M <= 17, "one 16x16 tile plus one border row");
    constexpr int MC = M < 16 ? M : 16, RT = (MC + 1) / 2;
    constexpr bool BORDER = (M == 17);
    // (the lane id is laundered: the per-round 0 / 1 masks below are functions of it alone, and a caller's stage loop would otherwise have all of
    // them hoisted out and kept live across its whole body -- 60+ registers, spilled)
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    mfma_d4 acc;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = lk + 4 * r, hi = max(row, li), lo = min(row, li);
        const double v = Y[min(hi, MC - 1) * ld + min(lo, MC - 1)];
        acc[r] = (hi < MC) ? v : ((row == li) ? 1.0 : 0.0);           // identity in the padding of a tile that is not full
    }
    double bd = 0.0, corner = 0.0;                                      // row 16: entry (16, li) on every lane; the corner (16, 16)
    if constexpr (BORDER) { bd = Y[16 * ld + li]; corner = Y[16 * ld + 16]; }
    // selections as products with 0 / 1 (one instruction per double instead of two v_cndmask_b32; every value involved is finite)
    const double m0 = (lk == 0) ? 1.0 : 0.0, m1 = (lk == 1) ? 1.0 : 0.0;      // the lane groups that feed k = 0 / k = 1 of the rank-2 update
    double dmin = 1.0;            // smallest pivot seen (a pivot that is not positive is clamped; reported once at the end)
#pragma unroll
    for (int jb = 0; jb < RT; jb++) {
        const int pk0 = 2 * jb, pk1 = pk0 + 1, rp = pk0 / 4, g0 = pk0 % 4, g1 = pk1 % 4;      // both pivot rows sit in accumulator rp, lane groups g0, g1
        const double src = acc[rp];
        double c0 = bperm_d(src, li + 16 * g0), c1 = bperm_d(src, li + 16 * g1);                // rows pk0, pk1 at column li
        double e0 = 0.0, e1 = 0.0;
        if constexpr (BORDER) { e0 = readlane_d(bd, pk0); e1 = readlane_d(bd, pk1); }           // (16, pk0), (16, pk1)
        __builtin_amdgcn_sched_barrier(0);
        const double pa = readlane_d(src, pk0 + 16 * g0), pb = readlane_d(src, pk0 + 16 * g1), pc = readlane_d(src, pk1 + 16 * g1);
        const double d1 = fmax(pa, tiny_pivot<double>());                                        // (fmax also maps NaN to the clamp)
        const double r1 = fast_rcp(d1), bp = pb * r1;
        const double d2r = fma(-pb, bp, pc), d2 = fmax(d2r, tiny_pivot<double>());
        dmin = fmin(dmin, fmin(pa == pa ? pa : -1.0, d2r == d2r ? d2r : -1.0));
        const double r2 = fast_rcp(d2);
        const double q11 = r2, q01 = -bp * r2, q00 = fma(bp * bp, r2, r1);                      // P^-1 = [[q00, q01],[q01, q11]]
        const double j0 = (li == pk0) ? 1.0 : 0.0, j1 = (li == pk1) ? 1.0 : 0.0, keep = 1.0 - j0 - j1;     // column li in the pivot block?
        c0 = fma(c0, keep, -j0);                                                                // pivot rows / columns send -I
        c1 = fma(c1, keep, -j1);
        const double t0 = fma(c0, q00, c1 * q01), t1 = fma(c0, q01, c1 * q11);                  // row li of C P^-1
        const double av = fma(t0, m0, t1 * m1), bv = fma(c0, m0, c1 * m1);
        const double keepr = keep * (((lk == g0) | (lk == g1)) ? 0.0 : 1.0);                    // accumulator rp also holds the pivot ROWS
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] *= (r == rp) ? keepr : keep;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av, bv, acc, 0, 0, 0);
        if constexpr (BORDER) {
            const double u0 = fma(e0, q00, e1 * q01), u1 = fma(e0, q01, e1 * q11);              // (16, J) P^-1
            bd = fma(bd, keep, -fma(u0, c0, u1 * c1));
            corner -= fma(u0, e0, u1 * e1);
        }
    }
    if constexpr (BORDER) {      // last round: the 1x1 pivot (16, 16)
        const double d = fmax(corner, tiny_pivot<double>());
        dmin = fmin(dmin, corner == corner ? corner : -1.0);
        const double q = fast_rcp(d), t = bd * q;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-t * m0, bd * m0, acc, 0, 0, 0);
        bd = t; corner = -q;
    }
    // Dinv = -swept.  The two triangles of the tile were updated by separate MFMA lanes and differ in the last bit: the lower one is mirrored, so
    // that the matrix in LDS is exactly symmetric and equal to what the block-packed copy gives back
#pragma unroll
    for (int r = 0; r < 4; r++) { const int row = lk + 4 * r; if (row < MC && li <= row) { Dinv[row * ldi + li] = -acc[r]; Dinv[li * ldi + row] = -acc[r]; } }
    if constexpr (BORDER) {
        if (lk == 0) { Dinv[16 * ldi + li] = -bd; Dinv[li * ldi + 16] = -bd; }
        if (lane == 0) Dinv[16 * ldi + 16] = -corner;
    }
    wsync();
    if (blk != nullptr) {       // the block-packed copy for the solve-only sweeps (layout of spd_inv_gj / gj_blocks_to_lds)
        constexpr int NT = gj_blocks<M>();
        if (lane < NT) {
            const int bi = tri_row(lane), bj = lane - bi * (bi + 1) / 2, i0 = 2 * bi, i1 = min(i0 + 1, M - 1), l0 = 2 * bj, l1 = min(l0 + 1, M - 1);
            st2(blk + 4 * lane, d2{Dinv[i0 * ldi + l0], Dinv[i0 * ldi + l1]});
            st2(blk + 4 * lane + 2, d2{Dinv[i1 * ldi + l0], Dinv[i1 * ldi + l1]});
        }
    }
    return (dmin > tiny_pivot<double>()) ? 0 : 1;
}

// Solve the small SPD system H z = f (NU x NU, H in LDS broadcast) redundantly per lane; f/z in registers.
template <int NU>
__device__ __forceinline__ void spd_solve_small(const double *H, int ld, double *f) {
    double L[NU][NU], rd[NU];
#pragma unroll
    for (int i = 0; i < NU; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = H[i * ld + j];
#pragma unroll
            for (int k = 0; k < j; k++) s = fma(-L[i][k], L[j][k], s);
            L[i][j] = (i == j) ? s * fast_rsq(s) : s * rd[j];
            if (i == j) rd[j] = fast_rsq(s);
        }
#pragma unroll
    for (int i = 0; i < NU; i++) {
        double s = f[i];
#pragma unroll
        for (int k = 0; k < i; k++) s = fma(-L[i][k], f[k], s);
        f[i] = s * rd[i];
    }
#pragma unroll
    for (int i = NU - 1; i >= 0; i--) {
        double s = f[i];
#pragma unroll
        for (int k = i + 1; k < NU; k++) s = fma(-L[k][i], f[k], s);
        f[i] = s / L[i][i];
    }
}

}  // namespace wla
