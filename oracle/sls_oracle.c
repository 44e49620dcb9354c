/*
 * oracle/sls_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, not product code)
 *
 * Plain-C, single-instance, fp64 restatement of the reference's hot path
 * (antoineleeman/robust-nonlinear-mpc, citations relative to /root/reference):
 *
 *   so_backward   <- _backward_solve_numba + riccati_step_njit  solver/fast_SLS_jit.py:43-84
 *   so_propagate  <- _propagate                                  solver/fast_SLS_jit.py:87-117
 *   so_backoff    <- _backoff_from_phi                           solver/fast_SLS_jit.py:120-188
 *   so_qp_*       <- the QP that QP.solve hands to OSQP          solver/qp_jit.py:77-192 (row/col layout),
 *                                                                 :362-402 (x0 rows, BIG, status rule)
 *
 * Pinning status
 *   - sweep functions: PINNED against golden vectors produced by running the reference's own three
 *     kernels as plain NumPy (tests/golden/gen_golden.py, tests/test_oracle_sweep.py).
 *   - QP: the arithmetic lives in the third-party package osqp==1.0.4 (requirements.txt:31), which is
 *     not present in /root/reference nor in this image.  What follows restates OSQP's PUBLISHED algorithm
 *     (Stellato et al., "OSQP: an operator splitting solver for quadratic programs", Math. Prog. Comp.
 *     2020: Ruiz equilibration, ADMM with over-relaxation, rho_eq = 1e3 rho, adaptive rho, unscaled
 *     termination test, polish with regularised KKT + iterative refinement) from memory of that paper and
 *     the public documentation.  PARITY UNPINNED: there is no OSQP output anywhere in the reference to
 *     check against; the restatement is validated by KKT optimality certificates instead
 *     (tests/test_host_cpu.py: KKT certificates, unconstrained case against the reference's own Riccati recursion).  One deliberate implementation difference, mathematically neutral: the
 *     quasi-definite KKT system of each ADMM step is solved in its reduced form
 *     (P + sigma I + A' diag(rho) A) x = rhs with a banded Cholesky (stage ordering, half bandwidth
 *     nx+2nu+nx-1) instead of QDLDL + AMD.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef struct { int nx, nu, nw, N, ni, ni_f; } so_dims;

#define IDX2(i, j, ld) ((i) * (ld) + (j))

/* ------------------------------------------------------------------------------------------------
 * small dense helpers (row-major)
 * ---------------------------------------------------------------------------------------------- */
static void mm(int M, int N, int K, const double *A, int ta, const double *B, int tb, double *C) {
    /* C(MxN) = op(A) op(B); ta/tb: 1 = transposed storage (A is KxM when ta) */
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            double s = 0.0;
            for (int k = 0; k < K; k++) {
                double a = ta ? A[k * M + i] : A[i * K + k];
                double b = tb ? B[j * K + k] : B[k * N + j];
                s += a * b;
            }
            C[i * N + j] = s;
        }
}

/* Solve H Z = F for Z (H: n x n general, F: n x m) by partial-pivot LU, like np.linalg.solve. */
static int lu_solve(int n, int m, const double *H, const double *F, double *Z) {
    double *a = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    memcpy(a, H, sizeof(double) * n * n);
    memcpy(Z, F, sizeof(double) * n * m);
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++) if (fabs(a[r * n + c]) > fabs(a[p * n + c])) p = r;
        piv[c] = p;
        if (a[p * n + c] == 0.0) { free(a); free(piv); return 1; }
        if (p != c) {
            for (int j = 0; j < n; j++) { double t = a[c * n + j]; a[c * n + j] = a[p * n + j]; a[p * n + j] = t; }
            for (int j = 0; j < m; j++) { double t = Z[c * m + j]; Z[c * m + j] = Z[p * m + j]; Z[p * m + j] = t; }
        }
        for (int r = c + 1; r < n; r++) {
            double f = a[r * n + c] / a[c * n + c];
            a[r * n + c] = f;
            for (int j = c + 1; j < n; j++) a[r * n + j] -= f * a[c * n + j];
            for (int j = 0; j < m; j++) Z[r * m + j] -= f * Z[c * m + j];
        }
    }
    for (int r = n - 1; r >= 0; r--)
        for (int j = 0; j < m; j++) {
            double s = Z[r * m + j];
            for (int c = r + 1; c < n; c++) s -= a[r * n + c] * Z[c * m + j];
            Z[r * m + j] = s / a[r * n + r];
        }
    free(a); free(piv);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * SLS sweep  (fast_SLS_jit.py:43-188)
 * Shapes as in the reference: A(N,nx,nx) B(N,nx,nu) G(ni,nx+nu) Gf(ni_f,nx) eta(N,N,ni) eta_f(N+1,ni_f)
 * S(N+1,N+1,nx,nx) K(N,N+1,nu,nx) E(N+1,nx,nw) Phi_x(N+1,N+1,nx,nw) Phi_u(N,N+1,nu,nw)
 * beta(N,N,ni) beta_f(N+1,ni_f) backoff(N,ni) backoff_f(ni_f)
 * ---------------------------------------------------------------------------------------------- */
int so_backward(const so_dims *d, const double *A, const double *B, const double *G, const double *Gf,
                const double *eta, const double *eta_f, const double *Qreg, const double *Rreg,
                const double *Qregf, double *S, double *K) {
    const int nx = d->nx, nu = d->nu, N = d->N, ni = d->ni, nif = d->ni_f, nz = nx + nu;
    memset(S, 0, sizeof(double) * (N + 1) * (N + 1) * nx * nx);
    memset(K, 0, sizeof(double) * N * (N + 1) * nu * nx);
    double *C = (double *)malloc(sizeof(double) * nz * nz);
    double *x = (double *)malloc(sizeof(double) * nu * nx), *y = (double *)malloc(sizeof(double) * nx * nx);
    double *H = (double *)malloc(sizeof(double) * nu * nu), *F = (double *)malloc(sizeof(double) * nu * nx);
    double *Z = (double *)malloc(sizeof(double) * nu * nx), *Acl = (double *)malloc(sizeof(double) * nx * nx);
    double *Sn = (double *)malloc(sizeof(double) * nx * nx);
    int rc = 0;
    for (int jj = 0; jj <= N; jj++) {                                   /* :72 columns are independent */
        double *SN = S + ((size_t)(N * (N + 1) + jj)) * nx * nx;
        for (int a = 0; a < nx; a++)                                    /* :73-74 S[N,jj] = Gf' diag(eta_f) Gf + Q_reg_f */
            for (int b = 0; b < nx; b++) {
                double s = 0.0;
                for (int i = 0; i < nif; i++) s += Gf[i * nx + a] * eta_f[jj * nif + i] * Gf[i * nx + b];
                SN[a * nx + b] = s + Qregf[a * nx + b];
            }
        for (int kk = N - 1; kk >= jj; kk--) {                          /* :76 */
            const double *e = eta + ((size_t)(kk * N + jj)) * ni;
            for (int a = 0; a < nz; a++)                                /* :77 C = G' diag(eta) G */
                for (int b = 0; b < nz; b++) {
                    double s = 0.0;
                    for (int i = 0; i < ni; i++) s += G[i * nz + a] * e[i] * G[i * nz + b];
                    C[a * nz + b] = s;
                }
            const double *Ak = A + (size_t)kk * nx * nx, *Bk = B + (size_t)kk * nx * nu;
            const double *Sk = S + ((size_t)((kk + 1) * (N + 1) + jj)) * nx * nx;
            mm(nu, nx, nx, Bk, 1, Sk, 0, x);                            /* :47 x = B' S */
            mm(nx, nx, nx, Ak, 1, Sk, 0, y);                            /* :48 y = A' S */
            mm(nu, nu, nx, x, 0, Bk, 0, H);                             /* :50 H = Cu + x B */
            for (int a = 0; a < nu; a++)
                for (int b = 0; b < nu; b++) H[a * nu + b] += C[(nx + a) * nz + nx + b] + Rreg[a * nu + b];
            mm(nu, nx, nx, x, 0, Ak, 0, F);                             /* :51 F = x A */
            if (lu_solve(nu, nx, H, F, Z)) rc = 1;                      /* :54 */
            double *Kk = K + ((size_t)(kk * (N + 1) + jj)) * nu * nx;
            for (int i = 0; i < nu * nx; i++) Kk[i] = -Z[i];            /* :55 */
            mm(nx, nx, nu, Bk, 0, Kk, 0, Acl);                          /* :58 S = Cx + y (A + B K) */
            for (int i = 0; i < nx * nx; i++) Acl[i] += Ak[i];
            mm(nx, nx, nx, y, 0, Acl, 0, Sn);
            for (int a = 0; a < nx; a++)
                for (int b = 0; b < nx; b++) Sn[a * nx + b] += C[a * nz + b] + Qreg[a * nx + b];
            double *So = S + ((size_t)(kk * (N + 1) + jj)) * nx * nx;
            for (int a = 0; a < nx; a++)                                /* :61 symmetrise */
                for (int b = 0; b < nx; b++) So[a * nx + b] = 0.5 * (Sn[a * nx + b] + Sn[b * nx + a]);
        }
    }
    free(C); free(x); free(y); free(H); free(F); free(Z); free(Acl); free(Sn);
    return rc;
}

void so_propagate(const so_dims *d, const double *A, const double *B, const double *E, const double *K,
                  double *Phix, double *Phiu) {
    const int nx = d->nx, nu = d->nu, nw = d->nw, N = d->N;
    memset(Phix, 0, sizeof(double) * (N + 1) * (N + 1) * nx * nw);
    memset(Phiu, 0, sizeof(double) * N * (N + 1) * nu * nw);
    double *Acl = (double *)malloc(sizeof(double) * nx * nx);
    for (int j = 0; j <= N; j++)                                        /* :108-109 */
        memcpy(Phix + ((size_t)(j * (N + 1) + j)) * nx * nw, E + (size_t)j * nx * nw, sizeof(double) * nx * nw);
    for (int kk = 0; kk < N; kk++)
        for (int jj = 0; jj <= kk; jj++) {                              /* :113-116 */
            const double *Kk = K + ((size_t)(kk * (N + 1) + jj)) * nu * nx;
            const double *P = Phix + ((size_t)(kk * (N + 1) + jj)) * nx * nw;
            mm(nu, nw, nx, Kk, 0, P, 0, Phiu + ((size_t)(kk * (N + 1) + jj)) * nu * nw);
            mm(nx, nx, nu, B + (size_t)kk * nx * nu, 0, Kk, 0, Acl);
            for (int i = 0; i < nx * nx; i++) Acl[i] += A[(size_t)kk * nx * nx + i];
            mm(nx, nw, nx, Acl, 0, P, 0, Phix + ((size_t)((kk + 1) * (N + 1) + jj)) * nx * nw);
        }
    free(Acl);
}

void so_backoff(const so_dims *d, const double *Phix, const double *Phiu, const double *G, const double *Gf,
                double eps, double *beta, double *beta_f, double *backoff, double *backoff_f) {
    const int nx = d->nx, nu = d->nu, nw = d->nw, N = d->N, ni = d->ni, nif = d->ni_f, nz = nx + nu;
    memset(beta, 0, sizeof(double) * N * N * ni);
    for (int kk = 0; kk < N; kk++)
        for (int jj = 0; jj <= kk; jj++) {                              /* :144-158 */
            const double *Px = Phix + ((size_t)(kk * (N + 1) + jj)) * nx * nw;
            const double *Pu = Phiu + ((size_t)(kk * (N + 1) + jj)) * nu * nw;
            for (int i = 0; i < ni; i++) {
                double s = 0.0;
                for (int w = 0; w < nw; w++) {
                    double zx = 0.0, zu = 0.0;
                    for (int a = 0; a < nx; a++) zx += G[i * nz + a] * Px[a * nw + w];
                    for (int a = 0; a < nu; a++) zu += G[i * nz + nx + a] * Pu[a * nw + w];
                    double v = zx + zu;
                    s += v * v;
                }
                if (s < eps) s = eps;
                beta[((size_t)(kk * N + jj)) * ni + i] = s;
            }
        }
    for (int jj = 0; jj <= N; jj++) {                                   /* :161-170 */
        const double *Px = Phix + ((size_t)(N * (N + 1) + jj)) * nx * nw;
        for (int i = 0; i < nif; i++) {
            double s = 0.0;
            for (int w = 0; w < nw; w++) {
                double v = 0.0;
                for (int a = 0; a < nx; a++) v += Gf[i * nx + a] * Px[a * nw + w];
                s += v * v;
            }
            if (s < eps) s = eps;
            beta_f[jj * nif + i] = s;
        }
    }
    for (int kk = 0; kk < N; kk++)                                      /* :173-179 */
        for (int i = 0; i < ni; i++) {
            double acc = 0.0;
            for (int jj = 0; jj <= kk; jj++) acc += sqrt(beta[((size_t)(kk * N + jj)) * ni + i]);
            backoff[kk * ni + i] = acc;
        }
    for (int i = 0; i < nif; i++) {                                     /* :181-186 */
        double acc = 0.0;
        for (int jj = 0; jj <= N; jj++) acc += sqrt(beta_f[jj * nif + i]);
        backoff_f[i] = acc;
    }
}

/* ------------------------------------------------------------------------------------------------
 * QP data in the reference's layout (qp_jit.py:101-123, 178-186):
 *   variables y = [x0;u0;x1;u1;...;xN]            n = (nx+nu) N + nx
 *   rows      per stage k: nx rows [A_k B_k -I], ni rows G[x_k;u_k]; then ni_f rows Gf x_N; then nx rows I x_0
 *                                                   m = N (nx+ni) + ni_f + nx
 * A is kept as CSR (row-major sparse) built from the dense stage blocks; explicit zeros of A_k,B_k kept,
 * like the reference's frozen pattern.  P = 2*blkdiag(Q,R,...,Qf) (qp_jit.py:289: OSQP gets 2P).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int n, m, bw;           /* bw = half bandwidth of M = P + sigma I + A' rho A in stage ordering */
    int *rp, *ci;           /* CSR of A */
    double *av;             /* CSR values */
    double *P;              /* dense block-diagonal P stored as banded symmetric full n x (2*pb+1)?  -> we keep dense blocks */
    int nx, nu, N;
    double *Pblk;           /* (N+1) blocks of nz x nz (last nx x nx stored in nz x nz, zero padded) */
} so_qp;

static void qp_free(so_qp *q) { free(q->rp); free(q->ci); free(q->av); free(q->Pblk); }

static void qp_build(so_qp *q, const so_dims *d, const double *A, const double *B, const double *G,
                     const double *Gf, const double *Q, const double *R, const double *Qf) {
    const int nx = d->nx, nu = d->nu, N = d->N, ni = d->ni, nif = d->ni_f, nz = nx + nu;
    q->nx = nx; q->nu = nu; q->N = N;
    q->n = nz * N + nx;
    q->m = N * (nx + ni) + nif + nx;
    q->bw = nz + nx - 1;
    size_t nnz_max = (size_t)N * (nx * (nz + 1) + (size_t)ni * nz) + (size_t)nif * nx + nx;
    q->rp = (int *)malloc(sizeof(int) * (q->m + 1));
    q->ci = (int *)malloc(sizeof(int) * nnz_max);
    q->av = (double *)malloc(sizeof(double) * nnz_max);
    int r = 0, p = 0;
    for (int k = 0; k < N; k++) {
        int cx = k * nz, cu = cx + nx, cxp = (k + 1) * nz;
        for (int i = 0; i < nx; i++) {                                   /* qp_jit.py:105-113 */
            q->rp[r++] = p;
            for (int j = 0; j < nx; j++) { q->ci[p] = cx + j; q->av[p++] = A[((size_t)k * nx + i) * nx + j]; }
            for (int j = 0; j < nu; j++) { q->ci[p] = cu + j; q->av[p++] = B[((size_t)k * nx + i) * nu + j]; }
            q->ci[p] = cxp + i; q->av[p++] = -1.0;
        }
        for (int i = 0; i < ni; i++) {                                   /* :115-118 (sparse: zeros of G dropped) */
            q->rp[r++] = p;
            for (int j = 0; j < nz; j++)
                if (G[i * nz + j] != 0.0) { q->ci[p] = cx + j; q->av[p++] = G[i * nz + j]; }
        }
    }
    for (int i = 0; i < nif; i++) {                                      /* :120 */
        q->rp[r++] = p;
        for (int j = 0; j < nx; j++)
            if (Gf[i * nx + j] != 0.0) { q->ci[p] = N * nz + j; q->av[p++] = Gf[i * nx + j]; }
    }
    for (int i = 0; i < nx; i++) { q->rp[r++] = p; q->ci[p] = i; q->av[p++] = 1.0; }   /* :178-186 */
    q->rp[r] = p;
    q->Pblk = (double *)calloc((size_t)(N + 1) * nz * nz, sizeof(double));
    for (int k = 0; k <= N; k++) {
        double *Pb = q->Pblk + (size_t)k * nz * nz;
        const double *Qk = (k < N) ? Q : Qf;
        for (int a = 0; a < nx; a++) for (int b = 0; b < nx; b++) Pb[a * nz + b] = 2.0 * Qk[a * nx + b];
        if (k < N) for (int a = 0; a < nu; a++) for (int b = 0; b < nu; b++) Pb[(nx + a) * nz + nx + b] = 2.0 * R[a * nu + b];
    }
}

int so_qp_dims(const so_dims *d, int *n, int *m, int *nnzA) {
    const int nz = d->nx + d->nu;
    *n = nz * d->N + d->nx;
    *m = d->N * (d->nx + d->ni) + d->ni_f + d->nx;
    /* nnz with G=[I;-I], Gf=[I;-I] (352 / 5399 / 8371) */
    *nnzA = d->N * (d->nx * (nz + 1) + 2 * nz) + 2 * d->nx + d->nx;
    return 0;
}

/* y = A x ; y = A' x */
static void A_mul(const so_qp *q, const double *Av, const double *x, double *y) {
    for (int i = 0; i < q->m; i++) {
        double s = 0.0;
        for (int p = q->rp[i]; p < q->rp[i + 1]; p++) s += Av[p] * x[q->ci[p]];
        y[i] = s;
    }
}
static void At_mul(const so_qp *q, const double *Av, const double *v, double *y) {
    memset(y, 0, sizeof(double) * q->n);
    for (int i = 0; i < q->m; i++)
        for (int p = q->rp[i]; p < q->rp[i + 1]; p++) y[q->ci[p]] += Av[p] * v[i];
}
/* y = P x with P block-diagonal given by Ps (scaled copy of Pblk) */
static void P_mul(const so_qp *q, const double *Ps, const double *x, double *y) {
    const int nz = q->nx + q->nu;
    for (int k = 0; k <= q->N; k++) {
        int w = (k < q->N) ? nz : q->nx;
        const double *Pb = Ps + (size_t)k * nz * nz;
        for (int a = 0; a < w; a++) {
            double s = 0.0;
            for (int b = 0; b < w; b++) s += Pb[a * nz + b] * x[k * nz + b];
            y[k * nz + a] = s;
        }
    }
}
static double norm_inf(const double *v, int n) { double m = 0.0; for (int i = 0; i < n; i++) { double a = fabs(v[i]); if (a > m) m = a; } return m; }

/* banded symmetric positive definite matrix, lower band storage: Mb[i*(bw+1) + (bw + j - i)] for i-bw<=j<=i */
#define MB(i, j) Mb[(size_t)(i) * (bw + 1) + (bw + (j) - (i))]
static void band_assemble(const so_qp *q, const double *Av, const double *Ps, const double *rho, double sigma, double *Mb) {
    const int n = q->n, bw = q->bw, nz = q->nx + q->nu;
    memset(Mb, 0, sizeof(double) * (size_t)n * (bw + 1));
    for (int k = 0; k <= q->N; k++) {
        int w = (k < q->N) ? nz : q->nx;
        const double *Pb = Ps + (size_t)k * nz * nz;
        for (int a = 0; a < w; a++)
            for (int b = 0; b <= a; b++) MB(k * nz + a, k * nz + b) += Pb[a * nz + b];
    }
    for (int i = 0; i < n; i++) MB(i, i) += sigma;
    for (int r = 0; r < q->m; r++)
        for (int p1 = q->rp[r]; p1 < q->rp[r + 1]; p1++)
            for (int p2 = q->rp[r]; p2 < q->rp[r + 1]; p2++) {
                int i = q->ci[p1], j = q->ci[p2];
                if (j <= i) MB(i, j) += rho[r] * Av[p1] * Av[p2];
            }
}
static int band_chol(int n, int bw, double *Mb) {
    for (int j = 0; j < n; j++) {
        double dj = MB(j, j);
        int k0 = j - bw < 0 ? 0 : j - bw;
        for (int k = k0; k < j; k++) dj -= MB(j, k) * MB(j, k);
        if (!(dj > 0.0)) return 1;
        dj = sqrt(dj);
        MB(j, j) = dj;
        int i1 = j + bw >= n ? n - 1 : j + bw;
        for (int i = j + 1; i <= i1; i++) {
            double s = MB(i, j);
            int kk0 = i - bw < 0 ? 0 : i - bw;
            if (kk0 < k0) kk0 = k0;
            for (int k = kk0; k < j; k++) s -= MB(i, k) * MB(j, k);
            MB(i, j) = s / dj;
        }
    }
    return 0;
}
static void band_solve(int n, int bw, const double *Mb, double *x) {
    for (int i = 0; i < n; i++) {
        double s = x[i];
        int k0 = i - bw < 0 ? 0 : i - bw;
        for (int k = k0; k < i; k++) s -= MB(i, k) * x[k];
        x[i] = s / MB(i, i);
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        int k1 = i + bw >= n ? n - 1 : i + bw;
        for (int k = i + 1; k <= k1; k++) s -= MB(k, i) * x[k];
        x[i] = s / MB(i, i);
    }
}

/* OSQP settings (upstream defaults where known; see header comment) */
typedef struct {
    double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, delta;
    int max_iter, check_termination, scaling, adaptive_rho, adaptive_rho_interval, polish, polish_refine_iter;
    double adaptive_rho_tolerance;
} so_osqp_settings;

void so_osqp_default_settings(so_osqp_settings *s) {
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
    s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4; s->delta = 1e-6;
    s->max_iter = 4000; s->check_termination = 25; s->scaling = 10; s->adaptive_rho = 1;
    s->adaptive_rho_interval = 50; s->polish = 1; s->polish_refine_iter = 3; s->adaptive_rho_tolerance = 5.0;
}

typedef struct {
    int status;          /* 1 solved, 2 solved inaccurate(max iter but usable: not produced here), -3 primal infeasible, -4 dual infeasible, -2 max iter, -10 numerical */
    int iter, rho_updates, polish_status;   /* polish: 1 success, -1 unsuccessful, 0 not run */
    double obj_val, pri_res, dua_res, rho_final;
    double setup_time_ms, solve_time_ms;
} so_osqp_info;

#define OSQP_INFTY 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_TOL 1e-4
#define RHO_EQ_OVER_INEQ 1e3
#define RHO_MIN 1e-6
#define RHO_MAX 1e6

static double limit_scaling(double v) { if (v < MIN_SCALING) return 1.0; if (v > MAX_SCALING) return MAX_SCALING; return v; }

typedef struct {
    so_qp q;
    so_osqp_settings st;
    double *Av, *Ps;           /* scaled A values / P blocks */
    double *D, *E, *Dinv, *Einv; double c, cinv;
    double *rho_vec; int *ctype;      /* 1 eq, 0 ineq, -1 loose */
    double *Mb;
    double rho;
} so_work;

static void ruiz_scale(so_work *w, int iters) {
    so_qp *q = &w->q;
    const int n = q->n, m = q->m, nz = q->nx + q->nu;
    for (int i = 0; i < n; i++) w->D[i] = 1.0;
    for (int i = 0; i < m; i++) w->E[i] = 1.0;
    w->c = 1.0;
    double *Dt = (double *)malloc(sizeof(double) * n), *Et = (double *)malloc(sizeof(double) * m);
    for (int it = 0; it < iters; it++) {
        for (int i = 0; i < n; i++) Dt[i] = 0.0;
        for (int k = 0; k <= q->N; k++) {                 /* column inf-norms of P (symmetric) */
            int wd = (k < q->N) ? nz : q->nx;
            const double *Pb = w->Ps + (size_t)k * nz * nz;
            for (int a = 0; a < wd; a++) for (int b = 0; b < wd; b++) { double v = fabs(Pb[a * nz + b]); if (v > Dt[k * nz + b]) Dt[k * nz + b] = v; }
        }
        for (int r = 0; r < m; r++) {
            double rn = 0.0;
            for (int p = q->rp[r]; p < q->rp[r + 1]; p++) { double v = fabs(w->Av[p]); if (v > rn) rn = v; if (v > Dt[q->ci[p]]) Dt[q->ci[p]] = v; }
            Et[r] = rn;
        }
        for (int i = 0; i < n; i++) Dt[i] = 1.0 / sqrt(limit_scaling(Dt[i]));
        for (int r = 0; r < m; r++) Et[r] = 1.0 / sqrt(limit_scaling(Et[r]));
        for (int k = 0; k <= q->N; k++) {
            int wd = (k < q->N) ? nz : q->nx;
            double *Pb = w->Ps + (size_t)k * nz * nz;
            for (int a = 0; a < wd; a++) for (int b = 0; b < wd; b++) Pb[a * nz + b] *= Dt[k * nz + a] * Dt[k * nz + b];
        }
        for (int r = 0; r < m; r++) for (int p = q->rp[r]; p < q->rp[r + 1]; p++) w->Av[p] *= Et[r] * Dt[q->ci[p]];
        for (int i = 0; i < n; i++) w->D[i] *= Dt[i];
        for (int r = 0; r < m; r++) w->E[r] *= Et[r];
        /* cost normalisation; q == 0 at setup time in the reference (qp_jit.py:542) -> ||q|| limited to 1 */
        double mean = 0.0;
        for (int i = 0; i < n; i++) Dt[i] = 0.0;
        for (int k = 0; k <= q->N; k++) {
            int wd = (k < q->N) ? nz : q->nx;
            const double *Pb = w->Ps + (size_t)k * nz * nz;
            for (int a = 0; a < wd; a++) for (int b = 0; b < wd; b++) { double v = fabs(Pb[a * nz + b]); if (v > Dt[k * nz + b]) Dt[k * nz + b] = v; }
        }
        for (int i = 0; i < n; i++) mean += Dt[i];
        mean /= n;
        double ct = limit_scaling(mean);
        double qn = limit_scaling(0.0);
        if (qn > ct) ct = qn;
        ct = 1.0 / ct;
        for (size_t i = 0; i < (size_t)(q->N + 1) * nz * nz; i++) w->Ps[i] *= ct;
        w->c *= ct;
    }
    for (int i = 0; i < n; i++) w->Dinv[i] = 1.0 / w->D[i];
    for (int r = 0; r < m; r++) w->Einv[r] = 1.0 / w->E[r];
    w->cinv = 1.0 / w->c;
    free(Dt); free(Et);
}

static void set_rho_vec(so_work *w, const double *l, const double *u) {
    for (int i = 0; i < w->q.m; i++) {
        if (l[i] < -OSQP_INFTY * MIN_SCALING && u[i] > OSQP_INFTY * MIN_SCALING) { w->ctype[i] = -1; w->rho_vec[i] = RHO_MIN; }
        else if (u[i] - l[i] < RHO_TOL) { w->ctype[i] = 1; w->rho_vec[i] = RHO_EQ_OVER_INEQ * w->rho; }
        else { w->ctype[i] = 0; w->rho_vec[i] = w->rho; }
    }
}

/* polish: reduced-form regularised KKT with iterative refinement (OSQP paper, Sec. 5.2) */
static int polish(so_work *w, const double *qs, const double *l, const double *u, double *x, double *z, double *y,
                  double *pri_res, double *dua_res) {
    so_qp *q = &w->q;
    const int n = q->n, m = q->m, bw = q->bw;
    const double delta = w->st.delta;
    int *act = (int *)calloc(m, sizeof(int));
    double *b = (double *)calloc(m, sizeof(double)), *rw = (double *)calloc(m, sizeof(double));
    int nact = 0;
    for (int i = 0; i < m; i++) {
        if (z[i] - l[i] < -y[i]) { act[i] = -1; b[i] = l[i]; rw[i] = 1.0 / delta; nact++; }
        else if (u[i] - z[i] < y[i]) { act[i] = 1; b[i] = u[i]; rw[i] = 1.0 / delta; nact++; }
    }
    double *Mb = (double *)malloc(sizeof(double) * (size_t)n * (bw + 1));
    band_assemble(q, w->Av, w->Ps, rw, delta, Mb);
    int rc = band_chol(n, bw, Mb);
    double *xp = (double *)calloc(n, sizeof(double)), *yp = (double *)calloc(m, sizeof(double));
    double *r1 = (double *)malloc(sizeof(double) * n), *r2 = (double *)malloc(sizeof(double) * m);
    double *t1 = (double *)malloc(sizeof(double) * n), *t2 = (double *)malloc(sizeof(double) * m);
    if (!rc) {
        /* solve [P A_a'; A_a 0][x;y] = [-q; b_a] by refinement with the regularised matrix:
           (P + dI + A_a' A_a / d) dx = r1 + A_a' r2 / d ;  dy = (A_a dx - r2)/d */
        for (int it = 0; it <= w->st.polish_refine_iter; it++) {
            P_mul(q, w->Ps, xp, t1); At_mul(q, w->Av, yp, r1);
            for (int i = 0; i < n; i++) r1[i] = -qs[i] - t1[i] - r1[i];
            A_mul(q, w->Av, xp, t2);
            for (int i = 0; i < m; i++) r2[i] = act[i] ? (b[i] - t2[i]) : 0.0;
            for (int i = 0; i < m; i++) t2[i] = r2[i] * rw[i];
            At_mul(q, w->Av, t2, t1);
            for (int i = 0; i < n; i++) t1[i] += r1[i];
            band_solve(n, bw, Mb, t1);
            A_mul(q, w->Av, t1, t2);
            for (int i = 0; i < n; i++) xp[i] += t1[i];
            for (int i = 0; i < m; i++) if (act[i]) yp[i] += (t2[i] - r2[i]) * rw[i];
        }
        /* residuals of the polished point (scaled space is fine for the accept test: both sides scaled alike) */
        double *zp = (double *)malloc(sizeof(double) * m);
        A_mul(q, w->Av, xp, zp);
        double pr = 0.0;
        for (int i = 0; i < m; i++) { double v = 0.0; if (zp[i] < l[i]) v = l[i] - zp[i]; else if (zp[i] > u[i]) v = zp[i] - u[i]; v *= w->Einv[i]; if (v > pr) pr = v; }
        P_mul(q, w->Ps, xp, t1); At_mul(q, w->Av, yp, r1);
        double dr = 0.0;
        for (int i = 0; i < n; i++) { double v = fabs(t1[i] + qs[i] + r1[i]) * w->Dinv[i] * w->cinv; if (v > dr) dr = v; }
        int ok = (pr < *pri_res && dr < *dua_res) || (pr < *pri_res && *dua_res < 1e-10) || (dr < *dua_res && *pri_res < 1e-10);
        if (ok) {
            memcpy(x, xp, sizeof(double) * n); memcpy(y, yp, sizeof(double) * m);
            for (int i = 0; i < m; i++) z[i] = zp[i] < l[i] ? l[i] : (zp[i] > u[i] ? u[i] : zp[i]);
            *pri_res = pr; *dua_res = dr;
        }
        rc = ok ? 0 : 2;
        free(zp);
    }
    free(act); free(b); free(rw); free(Mb); free(xp); free(yp); free(r1); free(r2); free(t1); free(t2);
    return rc;    /* 0 success, 1 factorisation failed, 2 not better */
}

#include <time.h>
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

/*
 * One full "update_dynamics (OSQP setup) + solve" as the reference performs it per QP:
 *   setup  (qp_jit.py:537-548): scaling computed with q = 0, rho vector, factorisation
 *   update (qp_jit.py:391):     Ax (same values), q, l, u
 *   solve  (qp_jit.py:393)
 * l,u are in the reference row layout INCLUDING the trailing nx x0-rows; +-inf already mapped to +-1e20.
 * Outputs x (n), y (m) unscaled.
 */
int so_qp_solve(const so_dims *d, const double *A, const double *B, const double *G, const double *Gf,
                const double *Q, const double *R, const double *Qf, const double *qv, const double *lv,
                const double *uv, const so_osqp_settings *st, double *x_out, double *y_out, so_osqp_info *info) {
    so_work w;
    memset(&w, 0, sizeof(w));
    w.st = *st;
    double t0 = now_ms();
    qp_build(&w.q, d, A, B, G, Gf, Q, R, Qf);
    so_qp *q = &w.q;
    const int n = q->n, m = q->m, bw = q->bw, nz = q->nx + q->nu;
    const int nnz = q->rp[m];
    w.Av = (double *)malloc(sizeof(double) * nnz); memcpy(w.Av, q->av, sizeof(double) * nnz);
    w.Ps = (double *)malloc(sizeof(double) * (size_t)(q->N + 1) * nz * nz); memcpy(w.Ps, q->Pblk, sizeof(double) * (size_t)(q->N + 1) * nz * nz);
    w.D = (double *)malloc(sizeof(double) * n); w.Dinv = (double *)malloc(sizeof(double) * n);
    w.E = (double *)malloc(sizeof(double) * m); w.Einv = (double *)malloc(sizeof(double) * m);
    w.rho_vec = (double *)malloc(sizeof(double) * m); w.ctype = (int *)malloc(sizeof(int) * m);
    w.Mb = (double *)malloc(sizeof(double) * (size_t)n * (bw + 1));
    w.rho = st->rho;
    if (st->scaling > 0) ruiz_scale(&w, st->scaling);
    else { for (int i = 0; i < n; i++) w.D[i] = w.Dinv[i] = 1.0; for (int i = 0; i < m; i++) w.E[i] = w.Einv[i] = 1.0; w.c = w.cinv = 1.0; }
    double *qs = (double *)malloc(sizeof(double) * n), *l = (double *)malloc(sizeof(double) * m), *u = (double *)malloc(sizeof(double) * m);
    for (int i = 0; i < n; i++) qs[i] = w.c * w.D[i] * qv[i];
    for (int i = 0; i < m; i++) { l[i] = w.E[i] * lv[i]; u[i] = w.E[i] * uv[i]; }
    set_rho_vec(&w, l, u);
    band_assemble(q, w.Av, w.Ps, w.rho_vec, st->sigma, w.Mb);
    int rc = band_chol(n, bw, w.Mb);
    double t1 = now_ms();
    info->setup_time_ms = t1 - t0;
    info->rho_updates = 0; info->polish_status = 0;

    double *x = (double *)calloc(n, sizeof(double)), *z = (double *)calloc(m, sizeof(double)), *y = (double *)calloc(m, sizeof(double));
    double *xt = (double *)malloc(sizeof(double) * n), *zt = (double *)malloc(sizeof(double) * m);
    double *xp = (double *)malloc(sizeof(double) * n), *zp = (double *)malloc(sizeof(double) * m), *yp = (double *)malloc(sizeof(double) * m);
    double *tn = (double *)malloc(sizeof(double) * n), *tn2 = (double *)malloc(sizeof(double) * n), *tm = (double *)malloc(sizeof(double) * m);
    int status = -2, iter = 0;
    double pri = 0, dua = 0;
    if (rc) status = -10;
    for (iter = 1; !rc && iter <= st->max_iter; iter++) {
        memcpy(xp, x, sizeof(double) * n); memcpy(zp, z, sizeof(double) * m); memcpy(yp, y, sizeof(double) * m);
        /* rhs = sigma x - q + A'(rho z - y) */
        for (int i = 0; i < m; i++) tm[i] = w.rho_vec[i] * z[i] - y[i];
        At_mul(q, w.Av, tm, xt);
        for (int i = 0; i < n; i++) xt[i] += st->sigma * x[i] - qs[i];
        band_solve(n, bw, w.Mb, xt);
        A_mul(q, w.Av, xt, zt);
        for (int i = 0; i < n; i++) x[i] = st->alpha * xt[i] + (1.0 - st->alpha) * xp[i];
        for (int i = 0; i < m; i++) {
            double zr = st->alpha * zt[i] + (1.0 - st->alpha) * zp[i];
            double v = zr + y[i] / w.rho_vec[i];
            z[i] = v < l[i] ? l[i] : (v > u[i] ? u[i] : v);
            y[i] = y[i] + w.rho_vec[i] * (zr - z[i]);
        }
        int check = (st->check_termination > 0 && iter % st->check_termination == 0) || iter == st->max_iter;
        int adapt = st->adaptive_rho && st->adaptive_rho_interval > 0 && iter % st->adaptive_rho_interval == 0;
        if (!check && !adapt) continue;
        /* residuals (unscaled; scaled_termination = False is the upstream default) */
        A_mul(q, w.Av, x, tm);
        double nAx = 0, nz_ = 0, sAx = 0, sz = 0, spri = 0;
        pri = 0;
        for (int i = 0; i < m; i++) {
            double r = fabs(tm[i] - z[i]); if (r > spri) spri = r;
            r *= w.Einv[i]; if (r > pri) pri = r;
            double a = fabs(tm[i]); if (a > sAx) sAx = a; a *= w.Einv[i]; if (a > nAx) nAx = a;
            a = fabs(z[i]); if (a > sz) sz = a; a *= w.Einv[i]; if (a > nz_) nz_ = a;
        }
        P_mul(q, w.Ps, x, tn); At_mul(q, w.Av, y, tn2);
        double nPx = 0, nAty = 0, nq = 0, sPx = 0, sAty = 0, sq = 0, sdua = 0;
        dua = 0;
        for (int i = 0; i < n; i++) {
            double r = fabs(tn[i] + qs[i] + tn2[i]); if (r > sdua) sdua = r;
            r *= w.Dinv[i] * w.cinv; if (r > dua) dua = r;
            double a = fabs(tn[i]); if (a > sPx) sPx = a; a *= w.Dinv[i] * w.cinv; if (a > nPx) nPx = a;
            a = fabs(tn2[i]); if (a > sAty) sAty = a; a *= w.Dinv[i] * w.cinv; if (a > nAty) nAty = a;
            a = fabs(qs[i]); if (a > sq) sq = a; a *= w.Dinv[i] * w.cinv; if (a > nq) nq = a;
        }
        if (check) {
            double eps_p = st->eps_abs + st->eps_rel * (nAx > nz_ ? nAx : nz_);
            double md = nPx > nAty ? nPx : nAty; if (nq > md) md = nq;
            double eps_d = st->eps_abs + st->eps_rel * md;
            if (pri <= eps_p && dua <= eps_d) { status = 1; break; }
            /* primal infeasibility certificate on dy */
            double ndy = 0;
            for (int i = 0; i < m; i++) { tm[i] = y[i] - yp[i]; double a = fabs(tm[i]) * w.E[i]; if (a > ndy) ndy = a; }
            if (ndy > 1e-30) {
                double supp = 0.0;
                for (int i = 0; i < m; i++) {
                    double dyi = tm[i];
                    if (u[i] < OSQP_INFTY * MIN_SCALING && dyi > 0) supp += u[i] * dyi;
                    if (l[i] > -OSQP_INFTY * MIN_SCALING && dyi < 0) supp += l[i] * dyi;
                }
                At_mul(q, w.Av, tm, tn);
                double nAtdy = 0; for (int i = 0; i < n; i++) { double a = fabs(tn[i]) * w.Dinv[i]; if (a > nAtdy) nAtdy = a; }
                if (nAtdy <= st->eps_prim_inf * ndy && supp <= -st->eps_prim_inf * ndy) { status = -3; break; }
            }
        }
        if (adapt) {
            double a = sAx > sz ? sAx : sz, b = sPx > sAty ? sPx : sAty; if (sq > b) b = sq;
            double pn = spri / (a + 1e-10), dn = sdua / (b + 1e-10);
            double rn = w.rho * sqrt(pn / (dn + 1e-10));
            if (rn < RHO_MIN) rn = RHO_MIN; if (rn > RHO_MAX) rn = RHO_MAX;
            if (rn > w.rho * st->adaptive_rho_tolerance || rn < w.rho / st->adaptive_rho_tolerance) {
                w.rho = rn;
                for (int i = 0; i < m; i++) w.rho_vec[i] = w.ctype[i] == 1 ? RHO_EQ_OVER_INEQ * rn : (w.ctype[i] == -1 ? RHO_MIN : rn);
                band_assemble(q, w.Av, w.Ps, w.rho_vec, st->sigma, w.Mb);
                if (band_chol(n, bw, w.Mb)) { status = -10; break; }
                info->rho_updates++;
            }
        }
    }
    if (iter > st->max_iter) iter = st->max_iter;
    if (status == 1 && st->polish) {
        int pr = polish(&w, qs, l, u, x, z, y, &pri, &dua);
        info->polish_status = pr == 0 ? 1 : -1;
    }
    /* unscale */
    for (int i = 0; i < n; i++) x_out[i] = w.D[i] * x[i];
    for (int i = 0; i < m; i++) y_out[i] = w.cinv * w.E[i] * y[i];
    double obj = 0.0;
    P_mul(q, q->Pblk, x_out, tn);
    for (int i = 0; i < n; i++) obj += 0.5 * x_out[i] * tn[i] + qv[i] * x_out[i];
    info->status = status; info->iter = iter; info->obj_val = obj; info->pri_res = pri; info->dua_res = dua; info->rho_final = w.rho;
    info->solve_time_ms = now_ms() - t1;
    free(x); free(z); free(y); free(xt); free(zt); free(xp); free(zp); free(yp); free(tn); free(tn2); free(tm);
    free(qs); free(l); free(u);
    free(w.Av); free(w.Ps); free(w.D); free(w.Dinv); free(w.E); free(w.Einv); free(w.rho_vec); free(w.ctype); free(w.Mb);
    qp_free(q);
    return status == 1 ? 0 : 1;
}

/* KKT certificate of a primal-dual pair in the reference's row layout (used by tests for BOTH the oracle
 * and the HIP path).  Returns max of: stationarity ||Py+q+A'lam||inf, primal violation, dual sign violation,
 * complementarity |lam_i * dist-to-active-bound|. out[4]. */
void so_qp_kkt(const so_dims *d, const double *A, const double *B, const double *G, const double *Gf,
               const double *Q, const double *R, const double *Qf, const double *qv, const double *lv,
               const double *uv, const double *x, const double *y, double *out) {
    so_qp q;
    qp_build(&q, d, A, B, G, Gf, Q, R, Qf);
    double *t1 = (double *)malloc(sizeof(double) * q.n), *t2 = (double *)malloc(sizeof(double) * q.n), *z = (double *)malloc(sizeof(double) * q.m);
    P_mul(&q, q.Pblk, x, t1); At_mul(&q, q.av, y, t2);
    double st = 0, pv = 0, ds = 0, cp = 0;
    for (int i = 0; i < q.n; i++) { double r = fabs(t1[i] + qv[i] + t2[i]); if (r > st) st = r; }
    A_mul(&q, q.av, x, z);
    for (int i = 0; i < q.m; i++) {
        double v = 0; if (z[i] < lv[i]) v = lv[i] - z[i]; else if (z[i] > uv[i]) v = z[i] - uv[i]; if (v > pv) pv = v;
        double yp = y[i] > 0 ? y[i] : 0, ym = y[i] < 0 ? -y[i] : 0;
        double du = uv[i] - z[i], dl = z[i] - lv[i];
        double c = 0;
        if (uv[i] < 1e19) c = fabs(yp * du); else if (yp > ds) ds = yp;
        if (c > cp) cp = c;
        if (lv[i] > -1e19) c = fabs(ym * dl); else { c = 0; if (ym > ds) ds = ym; }
        if (c > cp) cp = c;
    }
    out[0] = st; out[1] = pv; out[2] = ds; out[3] = cp;
    free(t1); free(t2); free(z); qp_free(&q);
}

/* ------------------------------------------------------------------------------------------------
 * Batch driver for the CPU baseline of bench.py: one RTI fast-SLS step (fast_SLS.solve with rti_steps = 1,
 * solver/fast_SLS_jit.py:278-327: QP #1, evaluate_dual_eta, Riccati sweep, propagate, back-off, tightened QP #2)
 * for nb independent instances on nthreads POSIX threads, one instance per thread at a time -- the same
 * sequence oracle.py's OracleFastSLS.solve drives from Python, restated here so the all-core figure is not
 * limited by the interpreter lock.  Bounds as update_dynamics + offset_constraints (qp_jit.py:268-273, 595-610)
 * and update_tightening (fast_SLS_jit.py:556-569) set them, quirks q2/q3 included.
 * Inputs per instance b: A (N,nx,nx) B (N,nx,nu) g (N,ni) gN (ni_f) c (N,nx) q (n) x0 (nx); shared: G, Gf, gf_raw, E (N+1,nx,nw), weights.
 * Outputs: primal (nb,n) of the last QP, ok (nb): 1 = both QPs solved.
 * ---------------------------------------------------------------------------------------------- */
#include <pthread.h>
typedef struct {
    const so_dims *d; int nb; const double *A, *B, *g, *gN, *c, *q, *x0, *G, *Gf, *gf_raw, *E, *Q, *R, *Qf, *Qreg, *Rreg, *Qregf;
    const so_osqp_settings *st; double *primal; int *ok; int *next; pthread_mutex_t *mu; double budget_s, t0; int *done;
} rti_job;

static double wall_s(void) { return now_ms() * 1e-3; }

static void rti_one(const rti_job *J, int b) {
    const so_dims *d = J->d;
    const int nx = d->nx, nu = d->nu, nw = d->nw, N = d->N, ni = d->ni, nif = d->ni_f, nz = nx + nu, SR = nx + ni;
    const int n = nz * N + nx, mb = N * SR + nif, m = mb + nx;
    const double EPS = 1e-10, BIG = 1e20;
    const double *A = J->A + (size_t)b * N * nx * nx, *B = J->B + (size_t)b * N * nx * nu, *g = J->g + (size_t)b * N * ni,
                 *gN = J->gN + (size_t)b * nif, *c = J->c + (size_t)b * N * nx, *q = J->q + (size_t)b * n, *x0 = J->x0 + (size_t)b * nx;
    double *l = (double *)malloc(sizeof(double) * m), *u = (double *)malloc(sizeof(double) * m), *x = (double *)calloc(n, sizeof(double)),
           *y = (double *)calloc(m, sizeof(double));
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < nx; i++) { u[k * SR + i] = -c[k * nx + i] + EPS; l[k * SR + i] = -c[k * nx + i] - EPS; }
        for (int i = 0; i < ni; i++) { u[k * SR + nx + i] = g[k * ni + i] + EPS; l[k * SR + nx + i] = -BIG; }
    }
    for (int i = 0; i < nif; i++) { u[N * SR + i] = gN[i] + EPS; l[N * SR + i] = -BIG; }
    for (int i = 0; i < nx; i++) { u[mb + i] = -x0[i] + EPS; l[mb + i] = -x0[i] - EPS; }
    so_osqp_info info;
    int ok = 0;
    so_qp_solve(d, A, B, J->G, J->Gf, J->Q, J->R, J->Qf, q, l, u, J->st, x, y, &info);
    if (info.status == 1 || info.status == 2) {
        double *eta = (double *)calloc((size_t)N * N * ni, sizeof(double)), *eta_f = (double *)calloc((size_t)(N + 1) * nif, sizeof(double));
        const double s0 = 2.0 * sqrt(EPS);                                  /* beta = eps everywhere after initialize_backoff */
        for (int kk = 0; kk < N; kk++) for (int jj = 0; jj <= kk; jj++) for (int i = 0; i < ni; i++) eta[((size_t)kk * N + jj) * ni + i] = y[kk * SR + nx + i] / s0;
        for (int jj = 0; jj <= N; jj++) for (int i = 0; i < nif; i++) eta_f[(size_t)jj * nif + i] = y[N * SR + i] / s0;
        double *S = (double *)malloc(sizeof(double) * (size_t)(N + 1) * (N + 1) * nx * nx), *K = (double *)calloc((size_t)N * (N + 1) * nu * nx, sizeof(double));
        double *Px = (double *)malloc(sizeof(double) * (size_t)(N + 1) * (N + 1) * nx * nw), *Pu = (double *)malloc(sizeof(double) * (size_t)N * (N + 1) * nu * nw);
        double *beta = (double *)malloc(sizeof(double) * (size_t)N * N * ni), *beta_f = (double *)malloc(sizeof(double) * (size_t)(N + 1) * nif);
        double *bo = (double *)malloc(sizeof(double) * (size_t)N * ni), *bof = (double *)malloc(sizeof(double) * nif);
        memset(S, 0, sizeof(double) * (size_t)(N + 1) * (N + 1) * nx * nx);
        so_backward(d, A, B, J->G, J->Gf, eta, eta_f, J->Qreg, J->Rreg, J->Qregf, S, K);
        so_propagate(d, A, B, J->E, K, Px, Pu);
        so_backoff(d, Px, Pu, J->G, J->Gf, EPS, beta, beta_f, bo, bof);
        for (int k = 0; k < N; k++) {
            for (int i = 0; i < nx; i++) u[k * SR + i] = -c[k * nx + i];                        /* no +eps (quirk q3) */
            for (int i = 0; i < ni; i++) u[k * SR + nx + i] = g[k * ni + i] - bo[k * ni + i];
        }
        for (int i = 0; i < nif; i++) u[N * SR + i] = J->gf_raw[i] - bof[i];                    /* raw gf (quirk q2) */
        so_qp_solve(d, A, B, J->G, J->Gf, J->Q, J->R, J->Qf, q, l, u, J->st, x, y, &info);
        ok = (info.status == 1 || info.status == 2);
        free(eta); free(eta_f); free(S); free(K); free(Px); free(Pu); free(beta); free(beta_f); free(bo); free(bof);
    }
    memcpy(J->primal + (size_t)b * n, x, sizeof(double) * n);
    J->ok[b] = ok;
    free(l); free(u); free(x); free(y);
}

static void *rti_worker(void *arg) {
    rti_job *J = (rti_job *)arg;
    for (;;) {
        pthread_mutex_lock(J->mu);
        int b = *J->next;
        const int stop = b >= J->nb || (J->budget_s > 0 && wall_s() - J->t0 > J->budget_s);
        if (!stop) *J->next = b + 1;
        pthread_mutex_unlock(J->mu);
        if (stop) return NULL;
        rti_one(J, b);
        pthread_mutex_lock(J->mu); *J->done += 1; pthread_mutex_unlock(J->mu);
    }
}

/* Returns the number of instances completed (instances are taken in order until nb are done or budget_s seconds have passed). */
int so_rti_step_batch(const so_dims *d, int nb, int nthreads, double budget_s, const double *A, const double *B, const double *g, const double *gN,
                      const double *c, const double *q, const double *x0, const double *G, const double *Gf, const double *gf_raw, const double *E,
                      const double *Q, const double *R, const double *Qf, const double *Qreg, const double *Rreg, const double *Qregf,
                      const so_osqp_settings *st, double *primal, int *ok) {
    pthread_mutex_t mu;
    pthread_mutex_init(&mu, NULL);
    int next = 0, done = 0;
    rti_job J = {d, nb, A, B, g, gN, c, q, x0, G, Gf, gf_raw, E, Q, R, Qf, Qreg, Rreg, Qregf, st, primal, ok, &next, &mu, budget_s, wall_s(), &done};
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, rti_worker, &J);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&mu);
    return done;
}
