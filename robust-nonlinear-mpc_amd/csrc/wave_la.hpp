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

// LDS-only ordering between lanes of the (single) wave of a workgroup.  NOT __syncthreads(): its workgroup fence
// also waits for every outstanding global load/store (vmcnt(0)), which cost ~86 % of the wave's lifetime in the first
// profile.  LDS instructions of one wave execute in order, so waiting for lgkmcnt(0) and stopping the compiler from
// moving memory operations across this point is all that is needed.
__device__ __forceinline__ void wsync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
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

// C(MxN) = alpha * op(A) op(B) + beta * C ; row-major, leading dims given; op = transpose when TA/TB.
// Each lane owns outputs o = lane, lane+64, ... ; with 1x2 register blocking along N when N is even-ish.
template <int M, int N, int K, bool TA, bool TB>
__device__ __forceinline__ void gemm(const double *A, int lda, const double *B, int ldb, double *C, int ldc,
                                     double alpha, double beta, int lane) {
    constexpr int TOT = M * N;
#pragma unroll
    for (int o0 = 0; o0 < TOT; o0 += 64) {
        const int o = o0 + lane;
        if (o < TOT) {
            const int i = o / N, j = o % N;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const double a = TA ? A[k * lda + i] : A[i * lda + k];
                const double b = TB ? B[j * ldb + k] : B[k * ldb + j];
                s = fma(a, b, s);
            }
            double r = alpha * s;
            if (beta != 0.0) r += beta * C[i * ldc + j];
            C[i * ldc + j] = r;
        }
    }
}


// C(MxN) = alpha * A(MxK) * B(NxK)' with RBxCB register blocking per lane (one pass, needs ceil(M/RB)*ceil(N/CB) <= 64).
// Out-of-range rows/cols are clamped on load and masked on store.
template <int M, int N, int K, int RB, int CB>
__device__ __forceinline__ void gemm_nt_blk(const double *A, int lda, const double *B, int ldb, double *C, int ldc, double alpha, int lane) {
    constexpr int TR = (M + RB - 1) / RB, TC = (N + CB - 1) / CB;
    static_assert(TR * TC <= 64, "one pass only");
    if (lane < TR * TC) {
        const int tr = lane / TC, tc = lane % TC;
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
#pragma unroll
        for (int k = 0; k < K; k++) {
            double a[RB], b[CB];
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

// Lower triangle (i >= j) of  Y = M1 A' + B diag(piu) B' - L1 L1' + diag(d) ,  all NX x NX (B: NX x NU), 2x2 blocks, one pass.
// useL1 = false drops the L1 term.  Only the lower triangle of Y is written (the Cholesky reads nothing else).
template <int NX, int NU>
__device__ __forceinline__ void build_Y_lower(const double *M1, const double *A, const double *B, const double *piu, const double *L1,
                                              bool useL1, const double *d, double delta, double *Y, int lane) {
    constexpr int T = (NX + 1) / 2, NT = T * (T + 1) / 2;
    static_assert(NT <= 64, "one pass only");
    if (lane < NT) {
        // lane -> (bi >= bj) in the lower-triangular block grid
        int bi = 0, rem = lane;
        while (rem > bi) { rem -= bi + 1; bi++; }
        const int bj = rem;
        const int i0 = bi * 2, i1 = min(i0 + 1, NX - 1), j0 = bj * 2, j1 = min(j0 + 1, NX - 1);
        double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const double x0 = M1[i0 * NX + k], x1 = M1[i1 * NX + k], y0 = A[j0 * NX + k], y1 = A[j1 * NX + k];
            a00 = fma(x0, y0, a00); a01 = fma(x0, y1, a01); a10 = fma(x1, y0, a10); a11 = fma(x1, y1, a11);
        }
#pragma unroll
        for (int k = 0; k < NU; k++) {
            const double pk = piu[k];
            const double x0 = B[i0 * NU + k] * pk, x1 = B[i1 * NU + k] * pk, y0 = B[j0 * NU + k], y1 = B[j1 * NU + k];
            a00 = fma(x0, y0, a00); a01 = fma(x0, y1, a01); a10 = fma(x1, y0, a10); a11 = fma(x1, y1, a11);
        }
        if (useL1) {
#pragma unroll
            for (int k = 0; k < NX; k++) {
                const double x0 = L1[i0 * NX + k], x1 = L1[i1 * NX + k], y0 = L1[j0 * NX + k], y1 = L1[j1 * NX + k];
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

// y(M) = op(A)(MxK) x(K)  (lane i < M computes row i).  TA: A stored KxM.
template <int M, int K, bool TA>
__device__ __forceinline__ double matvec_row(const double *A, int lda, const double *x, int lane) {
    double s = 0.0;
    if (lane < M) {
#pragma unroll
        for (int k = 0; k < K; k++) s = fma(TA ? A[k * lda + lane] : A[lane * lda + k], x[k], s);
    }
    return s;
}

// In-place lower Cholesky of the MxM SPD matrix Y (LDS, ld), then Linv = L^{-1} (full MxM, upper part zero).
// Row i of Y lives in lane i's registers during the factorisation; column j is broadcast through `col`
// (LDS, >= M doubles).  Returns non-zero (wave-uniform) if a pivot was not positive (pivot clamped).
template <int M>
__device__ __forceinline__ int chol_inv(double *Y, int ld, double *Linv, int ldi, double *col, int lane) {
    double row[M];
    const bool act = lane < M;
#pragma unroll
    for (int c = 0; c < M; c++) row[c] = (act && c <= lane) ? Y[lane * ld + c] : 0.0;
    int fail = 0;
#pragma unroll
    for (int j = 0; j < M; j++) {
        // pivot (held by lane j) -> broadcast
        double d = __shfl(row[j], j);
        if (!(d > 1e-300)) { fail = 1; d = 1e-300; }
        const double rs = 1.0 / sqrt(d);
        if (act && lane >= j) row[j] = (lane == j) ? d * rs : row[j] * rs;
        if (act) col[lane] = row[j];
        wsync();
        if (act) {
#pragma unroll
            for (int l = j + 1; l < M; l++)
                if (lane >= l) row[l] = fma(-row[j], col[l], row[l]);
        }
        wsync();
    }
    // write L (lower) back, zero upper
    if (act) {
#pragma unroll
        for (int c = 0; c < M; c++) Y[lane * ld + c] = (c <= lane) ? row[c] : 0.0;
    }
    wsync();
    // inverse: lane c computes column c of X = L^{-1}
    double x[M];
#pragma unroll
    for (int i = 0; i < M; i++) {
        double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < i; m++) {
            // l_im is wave-uniform (LDS broadcast); x[m] is zero for m < c
            s = fma(-Y[i * ld + m], x[m], s);
        }
        const double xi = s / Y[i * ld + i];
        x[i] = (act && i >= lane) ? xi : 0.0;
    }
    if (act) {
#pragma unroll
        for (int i = 0; i < M; i++) Linv[i * ldi + lane] = x[i];
    }
    wsync();
    return fail;
}

// Solve the small SPD system H z = f (NU x NU, H in LDS broadcast) redundantly per lane; f/z in registers.
template <int NU>
__device__ __forceinline__ void spd_solve_small(const double *H, int ld, double *f) {
    double L[NU][NU];
#pragma unroll
    for (int i = 0; i < NU; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = H[i * ld + j];
#pragma unroll
            for (int k = 0; k < j; k++) s = fma(-L[i][k], L[j][k], s);
            L[i][j] = (i == j) ? sqrt(s) : s / L[j][j];
        }
#pragma unroll
    for (int i = 0; i < NU; i++) {
        double s = f[i];
#pragma unroll
        for (int k = 0; k < i; k++) s = fma(-L[i][k], f[k], s);
        f[i] = s / L[i][i];
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
