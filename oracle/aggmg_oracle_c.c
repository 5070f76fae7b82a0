/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (C restatement of the V-cycle hot path).
 *
 * Plain-C, single-thread restatement of the reference's hot path with the reference's
 * operation order, used (a) as a second checker beside oracle/aggmg_oracle.py and (b) as the
 * "port" CPU baseline bench.py times on the GPU box's host cores.  Never linked into or loaded
 * by the product (agglomerationmultigrid1d_amd).  Parity status: restatement-derived, reference
 * (Julia) not executed -- see the header of aggmg_oracle.py.
 *
 *   multigrid_v_cycle            src/solvers.jl:19-50
 *   apply_smoother(::BlockJacobi) src/smoother.jl:69-81   (per block LU solve, getrs order)
 *   apply_smoother(::JacobiSmoother) src/smoother.jl:56-58
 *   A*u (CSC scatter), L'*r (per-column dot), L*u (CSC scatter)   SparseArrays stdlib
 *   dg_smoother(:blockJac) block extraction + la.lu  src/smoother.jl:153-165 (getf2 order)
 *
 * Unlike the real reference this port allocates nothing inside the cycle (the Julia code makes
 * ~6 heap allocations and one LAPACK call per block per sweep), so it is a
 * faster CPU baseline than the reference itself would be.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
  int64_t n_rows, n_cols;
  const int64_t* colptr; /* 0-based */
  const int64_t* rowval; /* 0-based */
  const double* nzval;
} csc_t;

typedef struct {
  int kind;          /* 0 = point Jacobi, 1 = block Jacobi (contiguous blocks of size m) */
  int64_t m, nb;
  const double* diag; /* kind 0 */
  const double* lu;   /* kind 1: nb x m x m row-major LU factors (unit lower + upper) */
  const int32_t* piv; /* kind 1: nb x m pivot rows (0-based, getrf convention) */
} smoother_t;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* y = A*x : column scatter, each y[i] accumulated in ascending column order */
void oc_csc_matvec(const csc_t* A, const double* x, double* y) {
  memset(y, 0, sizeof(double) * (size_t)A->n_rows);
  for (int64_t j = 0; j < A->n_cols; ++j) {
    const double xj = x[j];
    for (int64_t p = A->colptr[j]; p < A->colptr[j + 1]; ++p) y[A->rowval[p]] += A->nzval[p] * xj;
  }
}

/* y = A'*x : one dot product per column, ascending row order */
void oc_csc_adjoint_matvec(const csc_t* A, const double* x, double* y) {
  for (int64_t j = 0; j < A->n_cols; ++j) {
    double acc = 0.0;
    for (int64_t p = A->colptr[j]; p < A->colptr[j + 1]; ++p) acc += A->nzval[p] * x[A->rowval[p]];
    y[j] = acc;
  }
}

/* LU with partial pivoting of one m x m row-major block (getf2 arithmetic). 0 ok, 1 singular */
int oc_block_lu(int m, double* a, int32_t* piv) {
  for (int k = 0; k < m; ++k) {
    int p = k;
    double best = fabs(a[k * m + k]);
    for (int i = k + 1; i < m; ++i)
      if (fabs(a[i * m + k]) > best) { best = fabs(a[i * m + k]); p = i; }
    piv[k] = p;
    if (a[p * m + k] == 0.0) return 1;
    if (p != k)
      for (int j = 0; j < m; ++j) { double t = a[k * m + j]; a[k * m + j] = a[p * m + j]; a[p * m + j] = t; }
    const double rp = 1.0 / a[k * m + k];
    for (int i = k + 1; i < m; ++i) a[i * m + k] *= rp;
    for (int i = k + 1; i < m; ++i) {
      const double l = a[i * m + k];
      for (int j = k + 1; j < m; ++j) a[i * m + j] -= l * a[k * m + j];
    }
  }
  return 0;
}

static void block_lu_solve(int m, const double* a, const int32_t* piv, double* x) {
  for (int k = 0; k < m; ++k)
    if (piv[k] != k) { double t = x[k]; x[k] = x[piv[k]]; x[piv[k]] = t; }
  for (int i = 1; i < m; ++i) { double s = x[i]; for (int j = 0; j < i; ++j) s -= a[i * m + j] * x[j]; x[i] = s; }
  for (int i = m - 1; i >= 0; --i) {
    double s = x[i];
    for (int j = i + 1; j < m; ++j) s -= a[i * m + j] * x[j];
    x[i] = s / a[i * m + i];
  }
}

/* dg_smoother(mesh, A, :blockJac): blocks[i] = lu(Matrix(A[inds_i, inds_i])), contiguous inds.
 * lu: nb*m*m, piv: nb*m.  Returns 0, or (1 + index of the singular block). */
int64_t oc_extract_factor_blocks(const csc_t* A, int64_t m, int64_t nb, double* lu, int32_t* piv) {
  memset(lu, 0, sizeof(double) * (size_t)(nb * m * m));
  for (int64_t j = 0; j < A->n_cols; ++j) {
    const int64_t blk = j / m;
    if (blk >= nb) break;
    for (int64_t p = A->colptr[j]; p < A->colptr[j + 1]; ++p) {
      const int64_t r = A->rowval[p];
      if (r / m == blk) lu[(blk * m + (r - blk * m)) * m + (j - blk * m)] = A->nzval[p];
    }
  }
  for (int64_t k = 0; k < nb; ++k)
    if (oc_block_lu((int)m, lu + k * m * m, piv + k * m)) return k + 1;
  return 0;
}

/* Y = alpha * (S \ B) : apply_smoother */
void oc_apply_smoother(const smoother_t* S, const double* B, double alpha, double* Y, int64_t N) {
  if (S->kind == 0) {
    for (int64_t i = 0; i < N; ++i) Y[i] = alpha * (B[i] / S->diag[i]);
    return;
  }
  const int m = (int)S->m;
  double tmp[64];
  memset(Y, 0, sizeof(double) * (size_t)N);
  for (int64_t k = 0; k < S->nb; ++k) {
    for (int i = 0; i < m; ++i) tmp[i] = B[k * m + i];
    block_lu_solve(m, S->lu + k * m * m, S->piv + k * m, tmp);
    for (int i = 0; i < m; ++i) Y[k * m + i] += tmp[i];
  }
  for (int64_t i = 0; i < N; ++i) Y[i] = alpha * Y[i];
}

/* one sweep: u += apply_smoother(S, rhs - A*u; alpha)   (src/solvers.jl:33) */
static void sweep(const csc_t* A, const smoother_t* S, const double* rhs, double alpha, double* u, double* t,
                  double* r, double* y) {
  const int64_t N = A->n_rows;
  oc_csc_matvec(A, u, t);
  for (int64_t i = 0; i < N; ++i) r[i] = rhs[i] - t[i];
  oc_apply_smoother(S, r, alpha, y, N);
  for (int64_t i = 0; i < N; ++i) u[i] = u[i] + y[i];
}

void oc_smooth(const csc_t* A, const smoother_t* S, const double* rhs, double alpha, int nsweeps, double* u,
               double* work /* 3*N */) {
  const int64_t N = A->n_rows;
  for (int s = 0; s < nsweeps; ++s) sweep(A, S, rhs, alpha, u, work, work + N, work + 2 * N);
}

/* banded LU (dgbtf2 / dgbtrs order) for the coarsest `A \ rhs` */
typedef struct {
  int64_t n;
  int kl, ku, ldab;
  double* ab;
  int32_t* ipiv;
} banded_t;

int oc_banded_factor(const csc_t* A, banded_t* f) {
  const int64_t n = A->n_rows;
  int kl = 0, ku = 0;
  for (int64_t j = 0; j < n; ++j)
    for (int64_t p = A->colptr[j]; p < A->colptr[j + 1]; ++p) {
      const int64_t i = A->rowval[p];
      if (i - j > kl) kl = (int)(i - j);
      if (j - i > ku) ku = (int)(j - i);
    }
  const int ldab = 2 * kl + ku + 1, kv = ku + kl;
  f->n = n; f->kl = kl; f->ku = ku; f->ldab = ldab;
  f->ab = (double*)calloc((size_t)ldab * (size_t)n, sizeof(double));
  f->ipiv = (int32_t*)calloc((size_t)n, sizeof(int32_t));
  if (!f->ab || !f->ipiv) return 2;
#define AB(r, c) f->ab[(size_t)(c) * ldab + (r)]
  for (int64_t j = 0; j < n; ++j)
    for (int64_t p = A->colptr[j]; p < A->colptr[j + 1]; ++p) AB(kv + A->rowval[p] - j, j) = A->nzval[p];
  int64_t ju = 0;
  for (int64_t j = 0; j < n; ++j) {
    const int64_t km = (kl < n - 1 - j) ? kl : n - 1 - j;
    int64_t jp = 0;
    double best = fabs(AB(kv, j));
    for (int64_t i = 1; i <= km; ++i)
      if (fabs(AB(kv + i, j)) > best) { best = fabs(AB(kv + i, j)); jp = i; }
    f->ipiv[j] = (int32_t)(j + jp);
    if (AB(kv + jp, j) == 0.0) return 1;
    int64_t cand = j + ku + jp; if (cand > n - 1) cand = n - 1;
    if (cand > ju) ju = cand;
    if (jp != 0)
      for (int64_t c = j; c <= ju; ++c) { double t = AB(kv + jp - (c - j), c); AB(kv + jp - (c - j), c) = AB(kv - (c - j), c); AB(kv - (c - j), c) = t; }
    if (km > 0) {
      const double rp = 1.0 / AB(kv, j);
      for (int64_t i = 1; i <= km; ++i) AB(kv + i, j) *= rp;
      for (int64_t c = j + 1; c <= ju; ++c) {
        const double t = AB(kv - (c - j), c);
        if (t != 0.0) for (int64_t i = 1; i <= km; ++i) AB(kv + i - (c - j), c) -= AB(kv + i, j) * t;
      }
    }
  }
  return 0;
}

void oc_banded_solve(const banded_t* f, double* b) {
  const int64_t n = f->n; const int ldab = f->ldab, kl = f->kl, kv = f->ku + f->kl;
  if (kl > 0)
    for (int64_t j = 0; j < n - 1; ++j) {
      const int64_t lm = (kl < n - 1 - j) ? kl : n - 1 - j;
      const int64_t l = f->ipiv[j];
      if (l != j) { double t = b[l]; b[l] = b[j]; b[j] = t; }
      const double bj = b[j];
      for (int64_t i = 1; i <= lm; ++i) b[j + i] -= bj * AB(kv + i, j);
    }
  for (int64_t j = n - 1; j >= 0; --j) {
    b[j] /= AB(kv, j);
    const double bj = b[j];
    const int64_t lo = (j - kv > 0) ? j - kv : 0;
    for (int64_t i = lo; i < j; ++i) b[i] -= bj * AB(kv - (j - i), j);
  }
#undef AB
}

void oc_banded_free(banded_t* f) { free(f->ab); free(f->ipiv); f->ab = 0; f->ipiv = 0; }

/* multigrid_v_cycle(H, x0, b; nPre, nPost, alpha) -> x_out (src/solvers.jl:19-50).
 * A[k], S[k] (k < nlevels-1), L[k]: level k+1 -> k; coarse: factored coarsest operator.
 * work: caller-provided, >= sum_k 5*N_k doubles.  Returns seconds spent in the coarsest solve
 * through *coarse_s (the metric excludes it). */
int oc_vcycle(int nlevels, const csc_t* A, const smoother_t* S, const csc_t* L, const banded_t* coarse,
              const double* x0, const double* b, int nPre, int nPost, double alpha, double* x_out, double* work,
              double* coarse_s) {
  double* u[16]; double* rhs[16]; double* t[16]; double* r[16]; double* y[16];
  if (nlevels < 1 || nlevels > 16) return 1;
  double* w = work;
  for (int k = 0; k < nlevels; ++k) {
    const int64_t N = A[k].n_rows;
    u[k] = w; rhs[k] = w + N; t[k] = w + 2 * N; r[k] = w + 3 * N; y[k] = w + 4 * N;
    w += 5 * N;
  }
  const int64_t N0 = A[0].n_rows;
  memcpy(u[0], x0, sizeof(double) * (size_t)N0);     /* u[1] = x0 (not mutated) */
  memcpy(rhs[0], b, sizeof(double) * (size_t)N0);
  for (int k = 0; k < nlevels - 1; ++k) {
    const int64_t N = A[k].n_rows;
    if (k > 0) memset(u[k], 0, sizeof(double) * (size_t)N);
    for (int i = 0; i < nPre; ++i) sweep(&A[k], &S[k], rhs[k], alpha, u[k], t[k], r[k], y[k]);
    oc_csc_matvec(&A[k], u[k], t[k]);
    for (int64_t i = 0; i < N; ++i) r[k][i] = rhs[k][i] - t[k][i];
    oc_csc_adjoint_matvec(&L[k], r[k], rhs[k + 1]);
  }
  {
    const int k = nlevels - 1;
    const double t0 = now_s();
    memcpy(u[k], rhs[k], sizeof(double) * (size_t)A[k].n_rows);
    oc_banded_solve(coarse, u[k]);
    if (coarse_s) *coarse_s = now_s() - t0;
  }
  for (int k = nlevels - 2; k >= 0; --k) {
    const int64_t N = A[k].n_rows;
    oc_csc_matvec(&L[k], u[k + 1], t[k]);
    for (int64_t i = 0; i < N; ++i) u[k][i] = u[k][i] + t[k][i];
    for (int i = 0; i < nPost; ++i) sweep(&A[k], &S[k], rhs[k], alpha, u[k], t[k], r[k], y[k]);
  }
  memcpy(x_out, u[0], sizeof(double) * (size_t)N0);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * OpenMP variant of the same cycle ("best CPU" line of BASELINE.md section 5): row-parallel CSR
 * gathers instead of the serial CSC scatter (the per-row sums run over ascending columns, i.e.
 * the same order as the scatter accumulates them), blocks / columns / vector entries in
 * parallel.  Build with -fopenmp; without it the pragmas are ignored and this is a second serial
 * implementation.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t n_rows;
  const int64_t* rowptr;
  const int64_t* colind;
  const double* val;
} csr_t;

static void csr_residual(const csr_t* A, const double* u, const double* rhs, double* r) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A->n_rows; ++i) {
    double t = 0.0;
    for (int64_t p = A->rowptr[i]; p < A->rowptr[i + 1]; ++p) t += A->val[p] * u[A->colind[p]];
    r[i] = rhs[i] - t;
  }
}

static void smoother_update_omp(const smoother_t* S, const double* r, double alpha, double* u, int64_t N) {
  if (S->kind == 0) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) u[i] = u[i] + alpha * (r[i] / S->diag[i]);
    return;
  }
  const int m = (int)S->m;
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < S->nb; ++k) {
    double tmp[64];
    for (int i = 0; i < m; ++i) tmp[i] = r[k * m + i];
    block_lu_solve(m, S->lu + k * m * m, S->piv + k * m, tmp);
    for (int i = 0; i < m; ++i) u[k * m + i] = u[k * m + i] + alpha * tmp[i];
  }
}

int oc_vcycle_omp(int nlevels, const csr_t* A, const smoother_t* S, const csc_t* L, const csr_t* Lr,
                  const banded_t* coarse, const double* x0, const double* b, int nPre, int nPost, double alpha,
                  double* x_out, double* work, double* coarse_s) {
  double* u[16]; double* rhs[16]; double* r[16];
  if (nlevels < 1 || nlevels > 16) return 1;
  double* w = work;
  for (int k = 0; k < nlevels; ++k) {
    const int64_t N = A[k].n_rows;
    u[k] = w; rhs[k] = w + N; r[k] = w + 2 * N;
    w += 5 * N;
  }
  const int64_t N0 = A[0].n_rows;
  memcpy(u[0], x0, sizeof(double) * (size_t)N0);
  memcpy(rhs[0], b, sizeof(double) * (size_t)N0);
  for (int k = 0; k < nlevels - 1; ++k) {
    const int64_t N = A[k].n_rows;
    if (k > 0) memset(u[k], 0, sizeof(double) * (size_t)N);
    for (int i = 0; i < nPre; ++i) { csr_residual(&A[k], u[k], rhs[k], r[k]); smoother_update_omp(&S[k], r[k], alpha, u[k], N); }
    csr_residual(&A[k], u[k], rhs[k], r[k]);
    const csc_t* Lk = &L[k];
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < Lk->n_cols; ++j) {
      double acc = 0.0;
      for (int64_t p = Lk->colptr[j]; p < Lk->colptr[j + 1]; ++p) acc += Lk->nzval[p] * r[k][Lk->rowval[p]];
      rhs[k + 1][j] = acc;
    }
  }
  {
    const int k = nlevels - 1;
    const double t0 = now_s();
    memcpy(u[k], rhs[k], sizeof(double) * (size_t)A[k].n_rows);
    oc_banded_solve(coarse, u[k]);
    if (coarse_s) *coarse_s = now_s() - t0;
  }
  for (int k = nlevels - 2; k >= 0; --k) {
    const int64_t N = A[k].n_rows;
    const csr_t* P = &Lr[k];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
      double t = 0.0;
      for (int64_t p = P->rowptr[i]; p < P->rowptr[i + 1]; ++p) t += P->val[p] * u[k + 1][P->colind[p]];
      u[k][i] = u[k][i] + t;
    }
    for (int i = 0; i < nPost; ++i) { csr_residual(&A[k], u[k], rhs[k], r[k]); smoother_update_omp(&S[k], r[k], alpha, u[k], N); }
  }
  memcpy(x_out, u[0], sizeof(double) * (size_t)N0);
  return 0;
}
