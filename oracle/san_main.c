/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Sanitizer harness of the C restatement (sanitizers on the
 * CPU build): a stand-alone program around aggmg_oracle_c.c, built with -fsanitize=address,undefined (san_asan) and with
 * -fsanitize=thread (san_tsan, the OpenMP variant under ThreadSanitizer), driven by tests/test_sanitizers_cpu.py.
 *
 *   san_xxx <hierarchy.bin> <x_out.bin>
 *
 * hierarchy.bin (little-endian, written by the test from a hierarchy of the NumPy restatement):
 *   int64 nlevels
 *   per level k:      int64 n_rows, n_cols, nnz; int64 colptr[n_cols+1]; int64 rowval[nnz]; double nzval[nnz]     (A_k, CSC)
 *   per level k < n-1: the same for L_k
 *   int64 block_size[nlevels-1]   (0 = point Jacobi)
 *   double b[n_rows of A_0]
 * Runs two serial V(3,3) cycles (oc_vcycle) and two of the OpenMP variant (oc_vcycle_omp, CSR copies built here), checks
 * that the two agree to round-off, writes the serial iterate (x_out.bin: doubles) for the test to compare bit for bit
 * with the unsanitized library's.
 */
#include "aggmg_oracle_c.c"

#include <stdio.h>

static void* xmalloc(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
  return p;
}

static void rd(void* dst, size_t size, size_t cnt, FILE* f) {
  if (fread(dst, size, cnt, f) != cnt) { fprintf(stderr, "short read\n"); exit(2); }
}

static csc_t read_csc(FILE* f) {
  int64_t h[3];
  rd(h, sizeof(int64_t), 3, f);
  csc_t A;
  A.n_rows = h[0]; A.n_cols = h[1];
  int64_t* cp = xmalloc(sizeof(int64_t) * (size_t)(h[1] + 1));
  int64_t* rv = xmalloc(sizeof(int64_t) * (size_t)h[2]);
  double* nz = xmalloc(sizeof(double) * (size_t)h[2]);
  rd(cp, sizeof(int64_t), (size_t)(h[1] + 1), f);
  rd(rv, sizeof(int64_t), (size_t)h[2], f);
  rd(nz, sizeof(double), (size_t)h[2], f);
  A.colptr = cp; A.rowval = rv; A.nzval = nz;
  return A;
}

/* row-gather copy of a CSC matrix, columns ascending inside a row (counting sort by row) */
static csr_t to_csr(const csc_t* A) {
  const int64_t nnz = A->colptr[A->n_cols];
  int64_t* rp = xmalloc(sizeof(int64_t) * (size_t)(A->n_rows + 1));
  int64_t* ci = xmalloc(sizeof(int64_t) * (size_t)nnz);
  double* v = xmalloc(sizeof(double) * (size_t)nnz);
  memset(rp, 0, sizeof(int64_t) * (size_t)(A->n_rows + 1));
  for (int64_t p = 0; p < nnz; ++p) rp[A->rowval[p] + 1]++;
  for (int64_t i = 0; i < A->n_rows; ++i) rp[i + 1] += rp[i];
  int64_t* at = xmalloc(sizeof(int64_t) * (size_t)A->n_rows);
  memcpy(at, rp, sizeof(int64_t) * (size_t)A->n_rows);
  for (int64_t j = 0; j < A->n_cols; ++j)
    for (int64_t p = A->colptr[j]; p < A->colptr[j + 1]; ++p) {
      const int64_t q = at[A->rowval[p]]++;
      ci[q] = j; v[q] = A->nzval[p];
    }
  free(at);
  csr_t R;
  R.n_rows = A->n_rows; R.rowptr = rp; R.colind = ci; R.val = v;
  return R;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s hierarchy.bin x_out.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int64_t nl;
  rd(&nl, sizeof(int64_t), 1, f);
  if (nl < 1 || nl > 16) { fprintf(stderr, "bad level count\n"); return 2; }
  csc_t A[16], L[16];
  csr_t Ar[16], Lr[16];
  smoother_t S[16];
  int64_t bs[16];
  for (int k = 0; k < nl; ++k) A[k] = read_csc(f);
  for (int k = 0; k < nl - 1; ++k) L[k] = read_csc(f);
  rd(bs, sizeof(int64_t), (size_t)(nl - 1), f);
  const int64_t N0 = A[0].n_rows;
  double* b = xmalloc(sizeof(double) * (size_t)N0);
  rd(b, sizeof(double), (size_t)N0, f);
  fclose(f);
  /* self-test of the detectors (the test runs these once and expects a report: a clean run then means something) */
  const char* self = getenv("AGGMG_SAN_SELFTEST");
  if (self && !strcmp(self, "race")) {
    double acc = 0.0;
#pragma omp parallel for
    for (int64_t i = 0; i < 100000; ++i) acc += b[i % N0];   /* unsynchronised shared update: a data race */
    printf("selftest race: %g\n", acc);
    return 0;
  }
  if (self && !strcmp(self, "oob")) {
    volatile int64_t past = N0;
    printf("selftest oob: %g\n", b[past]);                   /* one element past the end of a heap block */
    return 0;
  }
  int64_t total = 0;
  for (int k = 0; k < nl; ++k) total += A[k].n_rows;
  for (int k = 0; k < nl - 1; ++k) {
    const int64_t N = A[k].n_rows;
    memset(&S[k], 0, sizeof(S[k]));
    if (bs[k] == 0) {
      double* d = xmalloc(sizeof(double) * (size_t)N);
      for (int64_t j = 0; j < N; ++j) {
        d[j] = 0.0;
        for (int64_t p = A[k].colptr[j]; p < A[k].colptr[j + 1]; ++p)
          if (A[k].rowval[p] == j) d[j] = A[k].nzval[p];
      }
      S[k].kind = 0; S[k].m = 1; S[k].nb = N; S[k].diag = d;
    } else {
      const int64_t m = bs[k], nb = N / m;
      double* lu = xmalloc(sizeof(double) * (size_t)(nb * m * m));
      int32_t* piv = xmalloc(sizeof(int32_t) * (size_t)(nb * m));
      const int64_t st = oc_extract_factor_blocks(&A[k], m, nb, lu, piv);
      if (st != 0) { fprintf(stderr, "singular block %lld\n", (long long)st); return 3; }
      S[k].kind = 1; S[k].m = m; S[k].nb = nb; S[k].lu = lu; S[k].piv = piv;
    }
  }
  banded_t coarse;
  memset(&coarse, 0, sizeof(coarse));
  if (oc_banded_factor(&A[nl - 1], &coarse) != 0) { fprintf(stderr, "coarsest operator singular\n"); return 3; }
  double* work = xmalloc(sizeof(double) * (size_t)(5 * total));
  double* x = xmalloc(sizeof(double) * (size_t)N0);
  double* y = xmalloc(sizeof(double) * (size_t)N0);
  double* xo = xmalloc(sizeof(double) * (size_t)N0);
  double* yo = xmalloc(sizeof(double) * (size_t)N0);
  memset(x, 0, sizeof(double) * (size_t)N0);
  memset(xo, 0, sizeof(double) * (size_t)N0);
  double cs = 0.0;
  for (int c = 0; c < 2; ++c) {           /* serial cycles, the reference's operation order */
    if (oc_vcycle((int)nl, A, S, L, &coarse, x, b, 3, 3, 2.0 / 3.0, y, work, &cs) != 0) return 4;
    double* t = x; x = y; y = t;
  }
  for (int k = 0; k < nl; ++k) Ar[k] = to_csr(&A[k]);
  for (int k = 0; k < nl - 1; ++k) Lr[k] = to_csr(&L[k]);
  for (int c = 0; c < 2; ++c) {           /* OpenMP row-gather variant */
    if (oc_vcycle_omp((int)nl, Ar, S, L, Lr, &coarse, xo, b, 3, 3, 2.0 / 3.0, yo, work, &cs) != 0) return 4;
    double* t = xo; xo = yo; yo = t;
  }
  double nd = 0.0, nx = 0.0;
  for (int64_t i = 0; i < N0; ++i) { nd += (x[i] - xo[i]) * (x[i] - xo[i]); nx += x[i] * x[i]; }
  if (!(sqrt(nd) <= 1e-9 * sqrt(nx))) { fprintf(stderr, "serial and OpenMP cycles differ: %g of %g\n", sqrt(nd), sqrt(nx)); return 5; }
  FILE* g = fopen(argv[2], "wb");
  if (!g || fwrite(x, sizeof(double), (size_t)N0, g) != (size_t)N0) { perror(argv[2]); return 2; }
  fclose(g);
  /* release everything: LeakSanitizer sees a clean exit */
  for (int k = 0; k < nl; ++k) {
    free((void*)A[k].colptr); free((void*)A[k].rowval); free((void*)A[k].nzval);
    free((void*)Ar[k].rowptr); free((void*)Ar[k].colind); free((void*)Ar[k].val);
  }
  for (int k = 0; k < nl - 1; ++k) {
    free((void*)L[k].colptr); free((void*)L[k].rowval); free((void*)L[k].nzval);
    free((void*)Lr[k].rowptr); free((void*)Lr[k].colind); free((void*)Lr[k].val);
    free((void*)S[k].diag); free((void*)S[k].lu); free((void*)S[k].piv);
  }
  oc_banded_free(&coarse);
  free(b); free(work); free(x); free(y); free(xo); free(yo);
  printf("san harness OK: %lld levels, N = %lld\n", (long long)nl, (long long)N0);
  return 0;
}
