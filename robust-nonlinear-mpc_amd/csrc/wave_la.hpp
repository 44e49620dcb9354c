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

__device__ __forceinline__ void wsync() { __syncthreads(); }

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
    for (int c = 0; c < M; c++) row[c] = act ? Y[(act ? lane : 0) * ld + c] : 0.0;
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
