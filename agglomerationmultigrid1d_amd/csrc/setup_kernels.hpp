// Set-up kernels of libaggmg_hip (gfx950, fp64): everything between "a SparseMatrixCSC arrived" and
// "the first V-cycle can be launched" runs on the device, straight from the uploaded CSC arrays --
//   validation + Int64 -> int32 conversion of the Julia arrays,
//   CSC -> CSR transposition (only for operators that end up on the generic kernels),
//   K6: batched block extraction + partial-pivot LU -> explicit inverse (dg_smoother(:blockJac),
//       src/smoother.jl:153-165; BlockDiagonalLU, src/block_diagonal.jl:47-58),
//   packing of the index-free block-tridiagonal / CG-chain forms, pattern and symmetry detection,
//   structured transfers, and the cyclic-reduction factorisation of the coarsest operator.
// This translation unit is compiled with -ffp-contract=off: the small dense factorisations then use
// the same IEEE operations, in the same order, as the host LU they replace (LAPACK getf2 order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aggmg {

constexpr int kSetupThreads = 256;

// ------------------------------------------------------------------------------------------
// upload: Julia CSC (Int64, 0- or 1-based) -> int32, validated
// ------------------------------------------------------------------------------------------
// err[0]: colptr not monotone / out of range, err[1]: row index out of range, err[2]: rows not strictly ascending
__global__ __launch_bounds__(kSetupThreads) void csc_convert_colptr_kernel(int64_t n, const int64_t* __restrict__ colptr64,
                                                                           int64_t base, int64_t nnz,
                                                                           int32_t* __restrict__ colptr,
                                                                           int* __restrict__ err) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j > n) return;
  const int64_t v = colptr64[j] - base;
  if (v < 0 || v > nnz || (j > 0 && v < colptr64[j - 1] - base)) err[0] = 1;
  colptr[j] = (int32_t)v;
}

__global__ __launch_bounds__(kSetupThreads) void csc_convert_rows_kernel(int64_t n, int64_t m,
                                                                         const int32_t* __restrict__ colptr,
                                                                         const int64_t* __restrict__ rowval64,
                                                                         int64_t base, int32_t* __restrict__ rowval,
                                                                         int* __restrict__ err) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j >= n) return;
  int64_t prev = -1;
  for (int32_t p = colptr[j]; p < colptr[j + 1]; ++p) {
    const int64_t r = rowval64[p] - base;
    if (r < 0 || r >= m) err[1] = 1;
    if (r <= prev) err[2] = 1;
    prev = r;
    rowval[p] = (int32_t)r;
  }
}

// column index of every stored entry (the expansion of colptr)
__global__ __launch_bounds__(kSetupThreads) void csc_entry_cols_kernel(int64_t n, const int32_t* __restrict__ colptr,
                                                                       int32_t* __restrict__ entry_col) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j >= n) return;
  for (int32_t p = colptr[j]; p < colptr[j + 1]; ++p) entry_col[p] = (int32_t)j;
}

__global__ __launch_bounds__(kSetupThreads) void iota_kernel(int64_t n, uint32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (i < n) out[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kSetupThreads) void row_count_kernel(int64_t nnz, const int32_t* __restrict__ rowval,
                                                                  int32_t* __restrict__ counts) {
  const int64_t p = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (p < nnz) atomicAdd(&counts[rowval[p]], 1);
}

// CSR arrays from the row-sorted entry permutation (stable: ascending column inside every row)
__global__ __launch_bounds__(kSetupThreads) void csr_gather_kernel(int64_t nnz, const uint32_t* __restrict__ perm,
                                                                   const int32_t* __restrict__ entry_col,
                                                                   const double* __restrict__ nzval,
                                                                   int32_t* __restrict__ colind,
                                                                   double* __restrict__ vals) {
  const int64_t q = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (q >= nnz) return;
  const uint32_t p = perm[q];
  colind[q] = entry_col[p];
  vals[q] = nzval[p];
}

// A[r, c] of a CSC matrix (rows ascending inside a column): 0.0 when not stored
__device__ __forceinline__ double csc_entry(const int32_t* __restrict__ colptr, const int32_t* __restrict__ rowval,
                                            const double* __restrict__ vals, int64_t r, int64_t c) {
  int32_t lo = colptr[c], hi = colptr[c + 1];
  while (lo < hi) {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if (rowval[mid] < r)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (lo < colptr[c + 1] && rowval[lo] == r) ? vals[lo] : 0.0;
}

// JacobiSmoother(Diagonal(A)) src/smoother.jl:95-102,146-152
__global__ __launch_bounds__(kSetupThreads) void diag_extract_kernel(int64_t n, const int32_t* __restrict__ colptr,
                                                                     const int32_t* __restrict__ rowval,
                                                                     const double* __restrict__ vals,
                                                                     double* __restrict__ diag) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j < n) diag[j] = csc_entry(colptr, rowval, vals, j, j);
}

// ------------------------------------------------------------------------------------------
// block index lists
// ------------------------------------------------------------------------------------------
// flags[0]: index out of range, flags[1]: not the contiguous aligned lists k*m + i, flags[2]: a node in two blocks
__global__ __launch_bounds__(kSetupThreads) void inds_convert_kernel(int64_t total, int64_t N, const int64_t* __restrict__ in,
                                                                     int64_t base, int32_t* __restrict__ out,
                                                                     double* __restrict__ counts, int* __restrict__ flags) {
  const int64_t t = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (t >= total) return;
  const int64_t v = in[t] - base;
  if (v < 0 || v >= N) {
    flags[0] = 1;
    out[t] = 0;
    return;
  }
  out[t] = (int32_t)v;
  if (v != t) flags[1] = 1;
  const double before = atomicAdd(&counts[v], 1.0);
  if (before > 0.0) flags[2] = 1;
}

// smallest index of every block's list; *unsorted set when the keys do not ascend with the block number
__global__ __launch_bounds__(kSetupThreads) void block_minkey_kernel(int64_t nb, int m, const int32_t* __restrict__ inds,
                                                                     uint32_t* __restrict__ keys, int* __restrict__ unsorted) {
  const int64_t k = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (k >= nb) return;
  int32_t mn = inds[k * m];
  for (int i = 1; i < m; ++i) mn = min(mn, inds[k * m + i]);
  keys[k] = (uint32_t)mn;
  if (k > 0) {
    int32_t pm = inds[(k - 1) * m];
    for (int i = 1; i < m; ++i) pm = min(pm, inds[(k - 1) * m + i]);
    if (pm > mn) *unsorted = 1;
  }
}

// blocks in a new order: new block k is old block order[k] (index lists and inverses)
__global__ __launch_bounds__(kSetupThreads) void block_permute_kernel(int64_t total, int m, const uint32_t* __restrict__ order,
                                                                      const int32_t* __restrict__ inds, const double* __restrict__ binv,
                                                                      int32_t* __restrict__ inds_out, double* __restrict__ binv_out) {
  const int64_t t = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;   // one thread per (block, row)
  if (t >= total) return;
  const int64_t k = t / m;
  const int i = (int)(t - k * m);
  const int64_t src = (int64_t)order[k] * m + i;
  inds_out[t] = inds[src];
  for (int j = 0; j < m; ++j) binv_out[t * m + j] = binv[src * m + j];
}

// rows [a, b] of a two-mode transfer: *flag != 0 unless every a is exactly 1.0; the b column on its own
__global__ __launch_bounds__(kSetupThreads) void transfer_unit_check_kernel(int64_t n, const double* __restrict__ lf, int* flag) {
  const int64_t i = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (i < n && lf[2 * i] != 1.0) *flag = 1;
}
__global__ __launch_bounds__(kSetupThreads) void transfer_second_column_kernel(int64_t n, const double* __restrict__ lf,
                                                                               double* __restrict__ lf1) {
  const int64_t i = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (i < n) lf1[i] = lf[2 * i + 1];
}

// dense blocks A[inds_k, inds_k] for arbitrary (possibly overlapping) index lists: one thread per entry
__global__ __launch_bounds__(kSetupThreads) void block_extract_generic_kernel(int64_t nb, int m,
                                                                              const int32_t* __restrict__ inds,
                                                                              const int32_t* __restrict__ colptr,
                                                                              const int32_t* __restrict__ rowval,
                                                                              const double* __restrict__ vals,
                                                                              double* __restrict__ blocks) {
  const int64_t t = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  const int64_t mm = (int64_t)m * m;
  if (t >= nb * mm) return;
  const int64_t k = t / mm;
  const int ij = (int)(t - k * mm);
  const int i = ij / m, j = ij - i * m;
  blocks[t] = csc_entry(colptr, rowval, vals, inds[k * m + i], inds[k * m + j]);
}

// ------------------------------------------------------------------------------------------
// K6: partial-pivot LU (LAPACK getf2 order) -> explicit inverse, one thread per block, in registers
// ------------------------------------------------------------------------------------------
// a: the block (row-major), destroyed; inv: its inverse.  false on an exactly-zero pivot
// (Julia: SingularException from la.lu, src/smoother.jl:160).
template <int M>
__device__ __forceinline__ bool lu_invert(double (&a)[M][M], double (&inv)[M][M]) {
  int piv[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    int p = k;
    double best = fabs(a[k][k]);
#pragma unroll
    for (int i = k + 1; i < M; ++i) {
      const double v = fabs(a[i][k]);
      if (v > best) {
        best = v;
        p = i;
      }
    }
    piv[k] = p;
    double pv = a[k][k];
#pragma unroll
    for (int i = k + 1; i < M; ++i)
      if (i == p) pv = a[i][k];
    if (pv == 0.0) return false;
#pragma unroll
    for (int i = k + 1; i < M; ++i)
      if (i == p) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const double t = a[k][j];
          a[k][j] = a[i][j];
          a[i][j] = t;
        }
      }
    const double rp = 1.0 / a[k][k];
#pragma unroll
    for (int i = k + 1; i < M; ++i) a[i][k] *= rp;
#pragma unroll
    for (int i = k + 1; i < M; ++i) {
      const double l = a[i][k];
#pragma unroll
      for (int j = k + 1; j < M; ++j) a[i][j] -= l * a[k][j];
    }
  }
#pragma unroll
  for (int c = 0; c < M; ++c) {
    double x[M];
#pragma unroll
    for (int i = 0; i < M; ++i) x[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
#pragma unroll
      for (int i = k + 1; i < M; ++i)
        if (i == piv[k]) {
          const double t = x[k];
          x[k] = x[i];
          x[i] = t;
        }
    }
#pragma unroll
    for (int i = 1; i < M; ++i) {
      double s = x[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= a[i][j] * x[j];
      x[i] = s;
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      double s = x[i];
#pragma unroll
      for (int j = i + 1; j < M; ++j) s -= a[i][j] * x[j];
      x[i] = s / a[i][i];
    }
#pragma unroll
    for (int i = 0; i < M; ++i) inv[i][c] = x[i];
  }
  return true;
}

// blocks [nb][M][M] row-major (or column-major when colmajor != 0: a Julia Matrix per block) -> inverses
// (row-major); singular[0] = smallest index of a singular block (initialised to nb)
template <int M>
__global__ __launch_bounds__(kSetupThreads) void block_invert_kernel(int64_t nb, const double* __restrict__ blocks,
                                                                     int colmajor, double* __restrict__ inv_out,
                                                                     unsigned long long* __restrict__ singular) {
  const int64_t k = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (k >= nb) return;
  double a[M][M], inv[M][M];
  const double* src = blocks + k * M * M;
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) a[i][j] = colmajor ? src[j * M + i] : src[i * M + j];
  if (!lu_invert<M>(a, inv)) {
    atomicMin(singular, (unsigned long long)k);
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) inv[i][j] = 0.0;
  }
  double* dst = inv_out + k * M * M;
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) dst[i * M + j] = inv[i][j];
}

// any block size up to 64: the same arithmetic with the work arrays in global memory (work: [nb][m*m + 2m])
__global__ __launch_bounds__(kSetupThreads) void block_invert_any_kernel(int64_t nb, int m, const double* __restrict__ blocks,
                                                                         int colmajor, double* __restrict__ work,
                                                                         double* __restrict__ inv_out,
                                                                         unsigned long long* __restrict__ singular) {
  const int64_t k = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (k >= nb) return;
  const int64_t mm = (int64_t)m * m;
  double* a = work + k * (mm + 2 * m);
  double* x = a + mm;
  double* pivf = x + m;
  const double* src = blocks + k * mm;
  double* inv = inv_out + k * mm;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      a[i * m + j] = colmajor ? src[j * m + i] : src[i * m + j];
      inv[i * m + j] = 0.0;
    }
  for (int kk = 0; kk < m; ++kk) {
    int p = kk;
    double best = fabs(a[kk * m + kk]);
    for (int i = kk + 1; i < m; ++i) {
      const double v = fabs(a[i * m + kk]);
      if (v > best) {
        best = v;
        p = i;
      }
    }
    pivf[kk] = (double)p;
    if (a[p * m + kk] == 0.0) {
      atomicMin(singular, (unsigned long long)k);
      return;
    }
    if (p != kk)
      for (int j = 0; j < m; ++j) {
        const double t = a[kk * m + j];
        a[kk * m + j] = a[p * m + j];
        a[p * m + j] = t;
      }
    const double rp = 1.0 / a[kk * m + kk];
    for (int i = kk + 1; i < m; ++i) a[i * m + kk] *= rp;
    for (int i = kk + 1; i < m; ++i) {
      const double l = a[i * m + kk];
      for (int j = kk + 1; j < m; ++j) a[i * m + j] -= l * a[kk * m + j];
    }
  }
  for (int c = 0; c < m; ++c) {
    for (int i = 0; i < m; ++i) x[i] = (i == c) ? 1.0 : 0.0;
    for (int kk = 0; kk < m; ++kk) {
      const int p = (int)pivf[kk];
      if (p != kk) {
        const double t = x[kk];
        x[kk] = x[p];
        x[p] = t;
      }
    }
    for (int i = 1; i < m; ++i) {
      double s = x[i];
      for (int j = 0; j < i; ++j) s -= a[i * m + j] * x[j];
      x[i] = s;
    }
    for (int i = m - 1; i >= 0; --i) {
      double s = x[i];
      for (int j = i + 1; j < m; ++j) s -= a[i * m + j] * x[j];
      x[i] = s / a[i * m + i];
    }
    for (int i = 0; i < m; ++i) inv[i * m + c] = x[i];
  }
}

// ------------------------------------------------------------------------------------------
// block-tridiagonal form
// ------------------------------------------------------------------------------------------
// Scatter every stored entry of A (CSC) into dblk / sub / sup [N][m] (zero-initialised).
// flags[0]: an entry outside the three block diagonals; masks[0]: columns of the sub-diagonal blocks that
// hold a non-zero, masks[1]: rows of the super-diagonal blocks that hold one (bit i = local index i).
__global__ __launch_bounds__(kSetupThreads) void btd_scatter_kernel(int64_t N, int m, const int32_t* __restrict__ colptr,
                                                                    const int32_t* __restrict__ rowval,
                                                                    const double* __restrict__ vals,
                                                                    double* __restrict__ dblk, double* __restrict__ sub,
                                                                    double* __restrict__ sup, int* __restrict__ flags,
                                                                    unsigned* __restrict__ masks) {
  const int64_t c = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (c >= N) return;
  const int64_t ce = c / m;
  const int cj = (int)(c - ce * m);
  for (int32_t p = colptr[c]; p < colptr[c + 1]; ++p) {
    const int64_t r = rowval[p];
    const int64_t e = r / m;
    const double v = vals[p];
    if (e == ce) {
      dblk[r * m + cj] = v;
    } else if (e == ce + 1) {  // A[block e, block e-1]
      sub[r * m + cj] = v;
      if (v != 0.0 && !(masks[0] & (1u << cj))) atomicOr(&masks[0], 1u << cj);
    } else if (e == ce - 1) {  // A[block e, block e+1]
      sup[r * m + cj] = v;
      const unsigned bit = 1u << (unsigned)(r - e * m);
      if (v != 0.0 && !(masks[1] & bit)) atomicOr(&masks[1], bit);
    } else {
      flags[0] = 1;
    }
  }
}

// compressed (nodal DG) pattern: scol = Sub[:, c_sub], qrow = Sup[r_sup, :], pcol = B^{-1} scol
__global__ __launch_bounds__(kSetupThreads) void btd_cmp_finish_kernel(int64_t ne, int m, int c_sub, int r_sup,
                                                                       const double* __restrict__ binv,
                                                                       const double* __restrict__ sub,
                                                                       const double* __restrict__ sup,
                                                                       double* __restrict__ scol, double* __restrict__ pcol,
                                                                       double* __restrict__ qrow) {
  const int64_t e = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (e >= ne) return;
  for (int j = 0; j < m; ++j) {
    qrow[e * m + j] = sup[(e * m + r_sup) * m + j];
    scol[e * m + j] = sub[(e * m + j) * m + c_sub];
  }
  for (int i = 0; i < m; ++i) {
    double acc = 0.0;
    for (int j = 0; j < m; ++j) acc += binv[(e * m + i) * m + j] * sub[(e * m + j) * m + c_sub];
    pcol[e * m + i] = acc;
  }
}

// dense pattern: P = B^{-1} Sub, Q = B^{-1} Sup
__global__ __launch_bounds__(kSetupThreads) void btd_dense_finish_kernel(int64_t ne, int m, const double* __restrict__ binv,
                                                                         const double* __restrict__ sub,
                                                                         const double* __restrict__ sup,
                                                                         double* __restrict__ P, double* __restrict__ Q) {
  const int64_t t = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (t >= ne * m) return;  // one thread per row (e, i)
  const int64_t e = t / m;
  for (int j = 0; j < m; ++j) {
    double p = 0.0, q = 0.0;
    for (int k = 0; k < m; ++k) {
      const double bi = binv[t * m + k];
      p += bi * sub[(e * m + k) * m + j];
      q += bi * sup[(e * m + k) * m + j];
    }
    P[t * m + j] = p;
    Q[t * m + j] = q;
  }
}

// symmetric to round-off?  asym[0] is raised when B_e^{-1} differs from its transpose, or Sub_e from Sup_{e-1}',
// by more than tol relative to the block's largest entry.  cmp != 0: only the compressed column / row are compared.
__global__ __launch_bounds__(kSetupThreads) void btd_sym_check_kernel(int64_t ne, int m, int cmp, int c_sub, int r_sup,
                                                                      double tol, const double* __restrict__ binv,
                                                                      const double* __restrict__ sub,
                                                                      const double* __restrict__ sup,
                                                                      int* __restrict__ asym) {
  const int64_t e = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (e >= ne) return;
  double scale = 0.0;
  for (int q = 0; q < m * m; ++q) scale = fmax(scale, fabs(binv[e * m * m + q]));
  bool bad = false;
  for (int i = 0; i < m; ++i)
    for (int j = i + 1; j < m; ++j)
      if (fabs(binv[(e * m + i) * m + j] - binv[(e * m + j) * m + i]) > tol * scale) bad = true;
  if (e > 0) {
    if (cmp) {
      double qs = 0.0;
      for (int j = 0; j < m; ++j) qs = fmax(qs, fabs(sup[((e - 1) * m + r_sup) * m + j]));
      for (int j = 0; j < m; ++j)
        if (fabs(sub[(e * m + j) * m + c_sub] - sup[((e - 1) * m + r_sup) * m + j]) > tol * qs) bad = true;
    } else {
      double ss = 0.0;
      for (int q = 0; q < m * m; ++q) ss = fmax(ss, fabs(sup[(e - 1) * m * m + q]));
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j)
          if (fabs(sub[(e * m + i) * m + j] - sup[((e - 1) * m + j) * m + i]) > tol * ss) bad = true;
    }
  }
  if (bad) asym[0] = 1;
}

// packed upper triangle of (B^{-1} + B^{-T}) / 2
__global__ __launch_bounds__(kSetupThreads) void btd_sym_pack_kernel(int64_t ne, int m, const double* __restrict__ binv,
                                                                     double* __restrict__ bsym) {
  const int64_t e = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (e >= ne) return;
  const int T = m * (m + 1) / 2;
  int q = 0;
  for (int i = 0; i < m; ++i)
    for (int j = i; j < m; ++j) bsym[e * T + q++] = 0.5 * (binv[(e * m + i) * m + j] + binv[(e * m + j) * m + i]);
}

// ------------------------------------------------------------------------------------------
// structured transfers of block-tridiagonal levels
// ------------------------------------------------------------------------------------------
// every stored entry of L (CSC, fine x coarse) must couple fine element e to the mc modes of coarse element
// e / rho: lf[(row)][mc] (zero-initialised); bad[0] raised otherwise
__global__ __launch_bounds__(kSetupThreads) void transfer_scatter_kernel(int64_t Nc, int mf, int mc, int64_t rho,
                                                                         const int32_t* __restrict__ colptr,
                                                                         const int32_t* __restrict__ rowval,
                                                                         const double* __restrict__ vals,
                                                                         double* __restrict__ lf, int* __restrict__ bad) {
  const int64_t c = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (c >= Nc) return;
  const int64_t J = c / mc;
  const int cj = (int)(c - J * mc);
  for (int32_t p = colptr[c]; p < colptr[c + 1]; ++p) {
    const int64_t r = rowval[p];
    if ((r / mf) / rho != J) {
      bad[0] = 1;
      return;
    }
    lf[r * mc + cj] = vals[p];
  }
}

// agglomerates of different sizes: every one of the mc columns of coarse element J stores the same contiguous run
// of fine rows [first[J] mf, first[J+1] mf), the runs of consecutive J follow each other and cover all fine rows.
// first[J], parent[e] and lf[(row)][mc]; bad[0] raised otherwise
__global__ __launch_bounds__(kSetupThreads) void transfer_vr_kernel(int64_t nec, int64_t nef, int mf, int mc, int maxch,
                                                                     const int32_t* __restrict__ colptr,
                                                                     const int32_t* __restrict__ rowval,
                                                                     const double* __restrict__ vals,
                                                                     int32_t* __restrict__ first, int32_t* __restrict__ parent,
                                                                     double* __restrict__ lf, int* __restrict__ bad) {
  const int64_t J = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (J >= nec) return;
  const int32_t p0 = colptr[J * mc];
  const int64_t len = colptr[J * mc + 1] - p0;
  if (len <= 0 || len % mf || len / mf > maxch) {
    bad[0] = 1;
    return;
  }
  const int64_t r0 = rowval[p0];
  if (r0 % mf) {
    bad[0] = 1;
    return;
  }
  // the run of the next coarse element must start where this one ends (the last one ends at the last fine row)
  const int64_t rnext = J + 1 < nec ? (int64_t)rowval[colptr[(J + 1) * mc]] : nef * mf;
  if ((J == 0 && r0 != 0) || r0 + len != rnext) {
    bad[0] = 1;
    return;
  }
  for (int c = 0; c < mc; ++c) {
    const int32_t pc = colptr[J * mc + c];
    if (colptr[J * mc + c + 1] - pc != len) {
      bad[0] = 1;
      return;
    }
    for (int64_t k = 0; k < len; ++k) {
      if (rowval[pc + k] != r0 + k) {
        bad[0] = 1;
        return;
      }
      lf[(r0 + k) * mc + c] = vals[pc + k];
    }
  }
  first[J] = (int32_t)(r0 / mf);
  atomicMax(&bad[1], (int)(len / mf));  // bad[1]: the largest agglomerate (fine elements)
  if (J == nec - 1) first[nec] = (int32_t)nef;
  for (int64_t e = r0 / mf; e < (r0 + len) / mf; ++e) parent[e] = (int32_t)J;
}

// ld[(e, j)][c] = sum_i lf[(e, i)][c] * D_e[i][j]   (rows of (L_e' D_e)')
__global__ __launch_bounds__(kSetupThreads) void transfer_ld_kernel(int64_t nef, int mf, int mc,
                                                                    const double* __restrict__ lf,
                                                                    const double* __restrict__ dblk,
                                                                    double* __restrict__ ld) {
  const int64_t t = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (t >= nef * mf) return;  // one thread per fine row (e, j)
  const int64_t e = t / mf;
  const int j = (int)(t - e * mf);
  for (int c = 0; c < mc; ++c) {
    double acc = 0.0;
    for (int i = 0; i < mf; ++i) acc += lf[(e * mf + i) * mc + c] * dblk[(e * mf + i) * mf + j];
    ld[t * mc + c] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// cyclic-reduction factorisation of the coarsest operator
// ------------------------------------------------------------------------------------------
// band[0] = max(i - j), band[1] = max(j - i) over the stored entries
__global__ __launch_bounds__(kSetupThreads) void band_kernel(int64_t n, const int32_t* __restrict__ colptr,
                                                             const int32_t* __restrict__ rowval, int* __restrict__ band) {
  const int64_t c = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (c >= n) return;
  const int32_t p0 = colptr[c], p1 = colptr[c + 1];
  if (p0 == p1) return;
  const int64_t lo = rowval[p0], hi = rowval[p1 - 1];
  if (hi > c) atomicMax(&band[0], (int)(hi - c));
  if (lo < c) atomicMax(&band[1], (int)(c - lo));
}

// does every entry stay within the three block diagonals for block size mm?  bad[0] raised if not
__global__ __launch_bounds__(kSetupThreads) void block_fit_kernel(int64_t n, int mm, const int32_t* __restrict__ colptr,
                                                                  const int32_t* __restrict__ rowval, int* __restrict__ bad) {
  const int64_t c = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (c >= n) return;
  const int64_t ce = c / mm;
  for (int32_t p = colptr[c]; p < colptr[c + 1]; ++p) {
    const int64_t e = rowval[p] / mm;
    if (e < ce - 1 || e > ce + 1) bad[0] = 1;
  }
}

// a, b, c [nblk][m][m] of the block rows (zero-initialised; b gets identity padding beyond N)
__global__ __launch_bounds__(kSetupThreads) void cr_pack_kernel(int64_t N, int m, const int32_t* __restrict__ colptr,
                                                                const int32_t* __restrict__ rowval,
                                                                const double* __restrict__ vals, double* __restrict__ a,
                                                                double* __restrict__ b, double* __restrict__ c) {
  const int64_t col = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  const int64_t nblk = (N + m - 1) / m;
  if (col >= nblk * m) return;
  const int64_t ce = col / m;
  const int lj = (int)(col - ce * m);
  if (col >= N) {
    b[ce * m * m + lj * m + lj] = 1.0;
    return;
  }
  for (int32_t p = colptr[col]; p < colptr[col + 1]; ++p) {
    const int64_t r = rowval[p];
    const int64_t e = r / m;
    const int li = (int)(r - e * m);
    double* dst = e == ce ? b : (e == ce + 1 ? a : c);  // row block below the column block: sub-diagonal a
    dst[e * m * m + li * m + lj] = vals[p];
  }
}

// parity-split storage of one level's off-diagonal blocks: fe[j] = (a_{2j}, c_{2j}) (turned into the forward multipliers
// by cr_even_multipliers_kernel), fo[j] = (a_{2j+1}, c_{2j+1})
__global__ __launch_bounds__(kSetupThreads) void cr_split_kernel(int64_t n, int mm2, const double* __restrict__ a,
                                                                 const double* __restrict__ c, double* __restrict__ fe,
                                                                 double* __restrict__ fo) {
  const int64_t t = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (t >= n * 2 * mm2) return;
  const int64_t r = t / (2 * mm2);
  const int k = (int)(t - r * 2 * mm2);
  double* dst = ((r & 1) ? fo : fe) + (r >> 1) * 2 * mm2 + k;
  *dst = k < mm2 ? a[r * mm2 + k] : c[r * mm2 + k - mm2];
}

__device__ __forceinline__ void atomic_max_pos(double* addr, double v) {
  // non-negative doubles order like their bit patterns
  atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

template <int M>
__device__ __forceinline__ double norm1_block(const double (&A)[M][M]) {
  double best = 0.0;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) s += fabs(A[i][j]);
    best = fmax(best, s);
  }
  return best;
}

// LU with partial pivoting stored as the factors of the row-permuted block (perm[k] = original row in
// position k), the form cr_lu_solve applies; false if singular
template <int M>
__device__ __forceinline__ bool lu_perm_dev(double (&lu)[M][M], int (&perm)[M]) {
#pragma unroll
  for (int k = 0; k < M; ++k) perm[k] = k;
#pragma unroll
  for (int k = 0; k < M; ++k) {
    int p = k;
    double best = fabs(lu[k][k]);
#pragma unroll
    for (int i = k + 1; i < M; ++i)
      if (fabs(lu[i][k]) > best) {
        best = fabs(lu[i][k]);
        p = i;
      }
    double pv = lu[k][k];
#pragma unroll
    for (int i = k + 1; i < M; ++i)
      if (i == p) pv = lu[i][k];
    if (pv == 0.0) return false;
#pragma unroll
    for (int i = k + 1; i < M; ++i)
      if (i == p) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const double t = lu[k][j];
          lu[k][j] = lu[i][j];
          lu[i][j] = t;
        }
        const int tp = perm[k];
        perm[k] = perm[i];
        perm[i] = tp;
      }
    const double rp = 1.0 / lu[k][k];
#pragma unroll
    for (int i = k + 1; i < M; ++i) lu[i][k] *= rp;
#pragma unroll
    for (int i = k + 1; i < M; ++i) {
      const double l = lu[i][k];
#pragma unroll
      for (int j = k + 1; j < M; ++j) lu[i][j] -= l * lu[k][j];
    }
  }
  return true;
}

// X = b \ R (column by column) with the permuted factors
template <int M>
__device__ __forceinline__ void lu_perm_solve_dev(const double (&lu)[M][M], const int (&perm)[M], const double (&R)[M][M],
                                                  double (&X)[M][M]) {
#pragma unroll
  for (int c = 0; c < M; ++c) {
    double y[M];
#pragma unroll
    for (int k = 0; k < M; ++k) {
      double v = R[0][c];
#pragma unroll
      for (int q = 1; q < M; ++q) v = (perm[k] == q) ? R[q][c] : v;
      y[k] = v;
    }
#pragma unroll
    for (int i = 1; i < M; ++i) {
      double s = y[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= lu[i][j] * y[j];
      y[i] = s;
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      double s = y[i];
#pragma unroll
      for (int j = i + 1; j < M; ++j) s -= lu[i][j] * y[j];
      y[i] = s / lu[i][i];
    }
#pragma unroll
    for (int k = 0; k < M; ++k) X[k][c] = y[k];
  }
}

// odd block rows of one reduction level: pivoted LU of b_{2j+1}, Za = b \ a, Zc = b \ c, condition monitor
template <int M>
__global__ __launch_bounds__(kSetupThreads) void cr_factor_odd_kernel(int64_t n_odd, const double* __restrict__ a,
                                                                      const double* __restrict__ b,
                                                                      const double* __restrict__ c,
                                                                      double* __restrict__ lu_out,
                                                                      int32_t* __restrict__ perm_out,
                                                                      double* __restrict__ Za, double* __restrict__ Zc,
                                                                      double* __restrict__ cond, int* __restrict__ bad) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j >= n_odd) return;
  const int64_t i = 2 * j + 1;
  double B[M][M], tb[M][M], tinv[M][M], A_[M][M], C_[M][M], X[M][M];
  int perm[M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int q = 0; q < M; ++q) {
      B[r][q] = b[i * M * M + r * M + q];
      tb[r][q] = B[r][q];
      A_[r][q] = a[i * M * M + r * M + q];
      C_[r][q] = c[i * M * M + r * M + q];
    }
  const double nb1 = norm1_block<M>(B);
  if (!lu_invert<M>(tb, tinv) || !lu_perm_dev<M>(B, perm)) {
    bad[0] = 1;
    return;
  }
  atomic_max_pos(cond, nb1 * norm1_block<M>(tinv));
#pragma unroll
  for (int r = 0; r < M; ++r) {
    perm_out[j * M + r] = perm[r];
#pragma unroll
    for (int q = 0; q < M; ++q) lu_out[j * M * M + r * M + q] = r == q ? 1.0 / B[r][r] : B[r][q];  // pivots as reciprocals
  }
  lu_perm_solve_dev<M>(B, perm, A_, X);
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int q = 0; q < M; ++q) Za[j * M * M + r * M + q] = X[r][q];
  lu_perm_solve_dev<M>(B, perm, C_, X);
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int q = 0; q < M; ++q) Zc[j * M * M + r * M + q] = X[r][q];
}

// The forward pass of the solve eliminates an odd row with its even neighbours' MULTIPLIERS a_{2j} b_{2j-1}^-1 and
// c_{2j} b_{2j+1}^-1 (row-vector solves with the pivoted factors  P b = L U:  a b^-1 = ((a U^-1) L^-1) P), formed here
// once -- the same block elimination the Schur complements above come from -- so that the per-cycle forward pass is
// multiply-adds on the even rows' blocks alone: it reads neither the LU factors nor the permutations and carries no
// triangular solves in its dependent chain.  One thread per odd row 2k+1 rewrites, in place, the c half of even row k
// and the a half of even row k+1 of fe (lu: unit-lower L, U with reciprocal pivots).
template <int M>
__global__ __launch_bounds__(kSetupThreads) void cr_even_multipliers_kernel(int64_t n_odd, int64_t n_even,
                                                                            const double* __restrict__ lu,
                                                                            const int32_t* __restrict__ perm,
                                                                            double* __restrict__ fe) {
  const int64_t k = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (k >= n_odd) return;
  double F[M][M];
  int pm[M];
#pragma unroll
  for (int r = 0; r < M; ++r) {
    pm[r] = perm[k * M + r];
#pragma unroll
    for (int q = 0; q < M; ++q) F[r][q] = lu[k * M * M + r * M + q];
  }
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int64_t row = k + side;  // side 0: c of even row k (right neighbour), side 1: a of even row k + 1 (left neighbour)
    if (row >= n_even) continue;
    double* blk = fe + row * 2 * M * M + (side == 0 ? M * M : 0);
#pragma unroll
    for (int r = 0; r < M; ++r) {
      double z[M], w[M];
#pragma unroll
      for (int q = 0; q < M; ++q) {  // z U = a_r
        double t = blk[r * M + q];
#pragma unroll
        for (int p2 = 0; p2 < q; ++p2) t -= z[p2] * F[p2][q];
        z[q] = t * F[q][q];
      }
#pragma unroll
      for (int q = M - 1; q >= 0; --q) {  // w L = z
        double t = z[q];
#pragma unroll
        for (int p2 = q + 1; p2 < M; ++p2) t -= w[p2] * F[p2][q];
        w[q] = t;
      }
      double out[M];
#pragma unroll
      for (int c2 = 0; c2 < M; ++c2) {  // (w P)_c = w_k for c = perm[k]
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < M; ++q) t = (pm[q] == c2) ? w[q] : t;
        out[c2] = t;
      }
#pragma unroll
      for (int c2 = 0; c2 < M; ++c2) blk[r * M + c2] = out[c2];
    }
  }
}

// ---- parallel cyclic reduction of the tail's boundary system (cr_kernels.hpp: cr_pcr_tail_kernel) ----------------------
// blk B^-1 by row-vector solves with the pivoted factors P B = L U (F: unit-lower L, U with RECIPROCAL pivots)
template <int M>
__device__ __forceinline__ void lu_rowsolve_dev(const double (&F)[M][M], const int (&pm)[M], const double (&blk)[M][M],
                                                double (&out)[M][M]) {
#pragma unroll
  for (int r = 0; r < M; ++r) {
    double z[M], w[M];
#pragma unroll
    for (int q = 0; q < M; ++q) {  // z U = blk_r
      double t = blk[r][q];
#pragma unroll
      for (int p2 = 0; p2 < q; ++p2) t -= z[p2] * F[p2][q];
      z[q] = t * F[q][q];
    }
#pragma unroll
    for (int q = M - 1; q >= 0; --q) {  // w L = z
      double t = z[q];
#pragma unroll
      for (int p2 = q + 1; p2 < M; ++p2) t -= w[p2] * F[p2][q];
      w[q] = t;
    }
#pragma unroll
    for (int c2 = 0; c2 < M; ++c2) {  // (w P)_c = w_k for c = perm[k]
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < M; ++q) t = (pm[q] == c2) ? w[q] : t;
      out[r][c2] = t;
    }
  }
}

// pivoted LU of every diagonal block of one PCR level (reciprocal pivots), bad[0] = 1 on a singular block
template <int M>
__global__ __launch_bounds__(kSetupThreads) void pcr_factor_kernel(int64_t n, const double* __restrict__ b,
                                                                   double* __restrict__ lu, int32_t* __restrict__ perm,
                                                                   int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (i >= n) return;
  double B[M][M];
  int pm[M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int q = 0; q < M; ++q) B[r][q] = b[i * M * M + r * M + q];
  if (!lu_perm_dev<M>(B, pm)) {
    bad[0] = 1;
    return;
  }
#pragma unroll
  for (int r = 0; r < M; ++r) {
    perm[i * M + r] = pm[r];
#pragma unroll
    for (int q = 0; q < M; ++q) lu[i * M * M + r * M + q] = r == q ? 1.0 / B[r][r] : B[r][q];
  }
}

// one PCR level at distance s: row i is rid of its couplings to rows i - s and i + s,
//   alpha = a_i b_{i-s}^-1, gamma = c_i b_{i+s}^-1  (kept: mult[i] = (alpha, gamma); zero where the neighbour does not exist)
//   a'_i = -alpha a_{i-s},  c'_i = -gamma c_{i+s},  b'_i = b_i - alpha c_{i-s} - gamma a_{i+s}   (couplings to i -+ 2s)
template <int M>
__global__ __launch_bounds__(kSetupThreads) void pcr_reduce_kernel(int64_t n, int64_t s, const double* __restrict__ a,
                                                                   const double* __restrict__ b, const double* __restrict__ c,
                                                                   const double* __restrict__ lu, const int32_t* __restrict__ perm,
                                                                   double* __restrict__ mult, double* __restrict__ a2,
                                                                   double* __restrict__ b2, double* __restrict__ c2) {
  const int64_t i = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (i >= n) return;
  constexpr int MM = M * M;
  double Bn[M][M], An[M][M], Cn[M][M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int q = 0; q < M; ++q) {
      Bn[r][q] = b[i * MM + r * M + q];
      An[r][q] = 0.0;
      Cn[r][q] = 0.0;
    }
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int64_t j = side == 0 ? i - s : i + s;
    double X[M][M];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int q = 0; q < M; ++q) X[r][q] = 0.0;
    if (j >= 0 && j < n) {
      double F[M][M], blk[M][M], aj[M][M], cj[M][M];
      int pm[M];
#pragma unroll
      for (int r = 0; r < M; ++r) {
        pm[r] = perm[j * M + r];
#pragma unroll
        for (int q = 0; q < M; ++q) {
          F[r][q] = lu[j * MM + r * M + q];
          blk[r][q] = (side == 0 ? a : c)[i * MM + r * M + q];
          aj[r][q] = a[j * MM + r * M + q];
          cj[r][q] = c[j * MM + r * M + q];
        }
      }
      lu_rowsolve_dev<M>(F, pm, blk, X);
#pragma unroll
      for (int r = 0; r < M; ++r)
#pragma unroll
        for (int q = 0; q < M; ++q) {
          double t1 = 0.0, t2 = 0.0;
#pragma unroll
          for (int k = 0; k < M; ++k) {
            t1 += X[r][k] * aj[k][q];
            t2 += X[r][k] * cj[k][q];
          }
          if (side == 0) {  // row i - s: its a couples to i - 2s, its c back to i
            An[r][q] = -t1;
            Bn[r][q] -= t2;
          } else {          // row i + s: its a couples back to i, its c to i + 2s
            Bn[r][q] -= t1;
            Cn[r][q] = -t2;
          }
        }
    }
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int q = 0; q < M; ++q) mult[i * 2 * MM + side * MM + r * M + q] = X[r][q];
  }
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int q = 0; q < M; ++q) {
      a2[i * MM + r * M + q] = An[r][q];
      b2[i * MM + r * M + q] = Bn[r][q];
      c2[i * MM + r * M + q] = Cn[r][q];
    }
}

// even block rows: Schur complement blocks of the next level
template <int M>
__global__ __launch_bounds__(kSetupThreads) void cr_schur_even_kernel(int64_t n, int64_t n_even, const double* __restrict__ a,
                                                                      const double* __restrict__ b,
                                                                      const double* __restrict__ c,
                                                                      const double* __restrict__ Za,
                                                                      const double* __restrict__ Zc, double* __restrict__ a2,
                                                                      double* __restrict__ b2, double* __restrict__ c2) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j >= n_even) return;
  const int64_t i = 2 * j;
  const int mm = M * M;
  double B2[M][M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int q = 0; q < M; ++q) {
      B2[r][q] = b[i * mm + r * M + q];
      a2[j * mm + r * M + q] = 0.0;
      c2[j * mm + r * M + q] = 0.0;
    }
  if (j > 0) {  // eliminate x_{i-1}: -a_i (b_{i-1} \ [a_{i-1} | c_{i-1}])
    const double* A = a + i * mm;
    const double* ZA = Za + (j - 1) * mm;
    const double* ZC = Zc + (j - 1) * mm;
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int q = 0; q < M; ++q) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
          s1 += A[r * M + k] * ZA[k * M + q];
          s2 += A[r * M + k] * ZC[k * M + q];
        }
        a2[j * mm + r * M + q] = -s1;
        B2[r][q] -= s2;
      }
  }
  if (i + 1 < n) {  // eliminate x_{i+1}: -c_i (b_{i+1} \ [a_{i+1} | c_{i+1}])
    const double* C = c + i * mm;
    const double* ZA = Za + j * mm;
    const double* ZC = Zc + j * mm;
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int q = 0; q < M; ++q) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
          s1 += C[r * M + k] * ZA[k * M + q];
          s2 += C[r * M + k] * ZC[k * M + q];
        }
        B2[r][q] -= s1;
        c2[j * mm + r * M + q] = -s2;
      }
  }
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int q = 0; q < M; ++q) b2[j * mm + r * M + q] = B2[r][q];
}

// the last remaining block: LU + condition monitor
template <int M>
__global__ void cr_factor_last_kernel(const double* __restrict__ b, double* __restrict__ lu_out,
                                      int32_t* __restrict__ perm_out, double* __restrict__ cond, int* __restrict__ bad) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double B[M][M], tb[M][M], tinv[M][M];
  int perm[M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int q = 0; q < M; ++q) {
      B[r][q] = b[r * M + q];
      tb[r][q] = B[r][q];
    }
  const double nb1 = norm1_block<M>(B);
  if (!lu_invert<M>(tb, tinv) || !lu_perm_dev<M>(B, perm)) {
    bad[0] = 1;
    return;
  }
  atomic_max_pos(cond, nb1 * norm1_block<M>(tinv));
#pragma unroll
  for (int r = 0; r < M; ++r) {
    perm_out[r] = perm[r];
#pragma unroll
    for (int q = 0; q < M; ++q) lu_out[r * M + q] = r == q ? 1.0 / B[r][r] : B[r][q];
  }
}

// mElements[e].mNodesInd of a CgMesh of degree m on nel elements in the reference's numbering (src/cg_mesh.jl:37-45,
// 59-65), 0-based: vertices e, e + 1, then the element's m - 1 interior nodes nel + 1 + e (m - 1) + i
__global__ __launch_bounds__(kSetupThreads) void chain_generate_elements_kernel(int64_t nel, int m, int64_t* __restrict__ elems) {
  const int64_t e = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (e >= nel) return;
  int64_t* o = elems + e * (m + 1);
  o[0] = e;
  o[1] = e + 1;
  for (int i = 0; i < m - 1; ++i) o[2 + i] = nel + 1 + e * (m - 1) + i;
}

// ------------------------------------------------------------------------------------------
// CG chain form (cgt_kernels.hpp)
// ------------------------------------------------------------------------------------------
// element lists ((p+1) x nel, column-major, Int64) -> block order: perm[e*m] = first node of element e,
// perm[e*m + j] = its node j + 1 (j >= 1); the second node must open the next element.
// flags[0]: index out of range (error), flags[1]: not a chain (generic path)
__global__ __launch_bounds__(kSetupThreads) void chain_perm_kernel(int64_t nel, int m, int64_t N, const int64_t* __restrict__ elems,
                                                                   int64_t base, int32_t* __restrict__ perm,
                                                                   int* __restrict__ flags) {
  const int64_t e = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (e >= nel) return;
  const int m1 = m + 1;
  for (int j = 0; j <= m; ++j) {
    const int64_t v = elems[e * m1 + j] - base;
    if (v < 0 || v >= N) {
      flags[0] = 1;
      return;
    }
  }
  if (e + 1 < nel && elems[e * m1 + 1] != elems[(e + 1) * m1]) flags[1] = 1;
  perm[e * m] = (int32_t)(elems[e * m1] - base);
  for (int j = 1; j < m; ++j) perm[e * m + j] = (int32_t)(elems[e * m1 + j + 1] - base);
  if (e == nel - 1) perm[nel * m] = (int32_t)(elems[e * m1 + 1] - base);
}

// rows of the element inverses for the Schwarz sweeps of the chain kernel: binv [nel][M+1][M+1] in the element
// lists' local order [left vertex, right vertex, interior ...] -> chain-local order [left vertex, interior ..., right
// vertex]; rows 0 .. M-1 of element e to zrows[(e*M + i)][M+1], its last row to zlast[e + 1][M+1]
__global__ __launch_bounds__(kSetupThreads) void chain_schwarz_rows_kernel(int64_t nel, int M, const double* __restrict__ binv,
                                                                           double* __restrict__ zrows, double* __restrict__ zlast) {
  const int64_t t = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  const int m1 = M + 1;
  if (t >= nel * m1) return;
  const int64_t e = t / m1;
  const int ci = (int)(t - e * m1);
  auto ref = [&](int c) { return c == 0 ? 0 : (c == M ? 1 : c + 1); };
  const double* src = binv + e * m1 * m1 + ref(ci) * m1;
  double* dst = ci < M ? zrows + (e * M + ci) * m1 : zlast + (e + 1) * m1;
  for (int cj = 0; cj <= M; ++cj) dst[cj] = src[ref(cj)];
}

// is the block order the reference's vertices-first numbering itself?  flags[3] raised if not
__global__ __launch_bounds__(kSetupThreads) void chain_affine_check_kernel(int64_t ne, int m, const int32_t* __restrict__ perm,
                                                                           int* __restrict__ flags) {
  const int64_t q = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (q >= ne * m) return;
  const int64_t e = q / m;
  const int i = (int)(q - e * m);
  const int64_t want = i == 0 ? e : (e == ne - 1 ? -1 : ne + e * (m - 1) + (i - 1));
  if (perm[q] != want) flags[3] = 1;
}

// inverse of a block order (entries -1 = padding); flags[1] raised when a node is listed twice
__global__ __launch_bounds__(kSetupThreads) void perm_invert_kernel(int64_t Np, const int32_t* __restrict__ perm,
                                                                    int32_t* __restrict__ inv, int* __restrict__ flags) {
  const int64_t q = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (q >= Np) return;
  const int32_t o = perm[q];
  if (o < 0) return;
  if (atomicCAS(&inv[o], -1, (int32_t)q) != -1) flags[1] = 1;
}

// flags[1] raised when a node of 0..N-1 is in no block
__global__ __launch_bounds__(kSetupThreads) void perm_cover_kernel(int64_t N, const int32_t* __restrict__ inv,
                                                                   int* __restrict__ flags) {
  const int64_t o = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (o < N && inv[o] < 0) flags[1] = 1;
}

// pack A (CSC, reference numbering) into the chain arrays; flags[2]: a non-zero outside the chain pattern
__global__ __launch_bounds__(kSetupThreads) void chain_scatter_kernel(int64_t N, int m, const int32_t* __restrict__ inv,
                                                                      const int32_t* __restrict__ colptr,
                                                                      const int32_t* __restrict__ rowval,
                                                                      const double* __restrict__ vals,
                                                                      double* __restrict__ dblk, double* __restrict__ subrow,
                                                                      double* __restrict__ supcol, int* __restrict__ flags) {
  const int64_t c = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (c >= N) return;
  const int64_t qc = inv[c];
  const int64_t ce = qc / m;
  const int cj = (int)(qc - ce * m);
  for (int32_t p = colptr[c]; p < colptr[c + 1]; ++p) {
    const int64_t q = inv[rowval[p]];
    const int64_t e = q / m;
    const int i = (int)(q - e * m);
    const double v = vals[p];
    if (ce == e)
      dblk[q * m + cj] = v;
    else if (ce == e - 1 && i == 0)
      subrow[e * m + cj] = v;
    else if (ce == e + 1 && cj == 0)
      supcol[q] = v;
    else if (v != 0.0)
      flags[2] = 1;
  }
}

// identity on the padding rows of the trailing block
__global__ __launch_bounds__(kSetupThreads) void chain_pad_kernel(int64_t Np, int m, const int32_t* __restrict__ perm,
                                                                  double* __restrict__ dblk) {
  const int64_t q = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (q < Np && perm[q] < 0) dblk[q * m + (q % m)] = 1.0;
}

// coarse block order read off a chain transfer L (CSC, fine x coarse) when the coarse level has no element
// lists: coarse column c with ONE... see cgt.hip; here: cvert[e] = the single column stored in fine vertex row
// perm_f[e*M] is found from the transposed side, so this kernel works per COARSE column: a coarse column
// whose first stored row is a fine vertex (block-order index divisible by M) is the vertex of that block.
// flags[1] raised on any inconsistency.
__global__ __launch_bounds__(kSetupThreads) void chain_coarse_vertices_kernel(int64_t Nc, int M, int mc,
                                                                              const int32_t* __restrict__ finv,
                                                                              const int32_t* __restrict__ colptr,
                                                                              const int32_t* __restrict__ rowval,
                                                                              int32_t* __restrict__ cperm,
                                                                              int32_t* __restrict__ cinv,
                                                                              int* __restrict__ flags) {
  const int64_t c = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (c >= Nc) return;
  // the fine rows of this coarse column, in block order: does it hold a fine vertex row?
  int64_t vertex_block = -1;
  for (int32_t p = colptr[c]; p < colptr[c + 1]; ++p) {
    const int64_t q = finv[rowval[p]];
    if (q % M == 0) {
      if (vertex_block >= 0) flags[1] = 1;  // two fine vertices interpolate from one coarse node
      vertex_block = q / M;
    }
  }
  if (vertex_block >= 0) {
    if (atomicCAS(&cperm[vertex_block * mc], -1, (int32_t)c) != -1) flags[1] = 1;
    cinv[c] = (int32_t)(vertex_block * mc);
  }
}

// coarse interior nodes of element e: the coarse columns (not vertices) that the first interior fine row of
// block e couples to, in ascending order -- one thread per coarse column decides its own slot: it belongs to the
// block of the fine rows it touches (all in one block) and its rank among that block's interior columns is the
// number of smaller interior columns touching the same block's first interior row
__global__ __launch_bounds__(kSetupThreads) void chain_coarse_interiors_kernel(int64_t Nc, int M, int mc,
                                                                               const int32_t* __restrict__ finv,
                                                                               const int32_t* __restrict__ colptr,
                                                                               const int32_t* __restrict__ rowval,
                                                                               int32_t* __restrict__ cperm,
                                                                               int32_t* __restrict__ cinv,
                                                                               int32_t* __restrict__ slot_counter,
                                                                               int* __restrict__ flags) {
  const int64_t c = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (c >= Nc || cinv[c] >= 0) return;  // vertices are placed already
  const int32_t p0 = colptr[c], p1 = colptr[c + 1];
  if (p0 == p1) {
    flags[1] = 1;
    return;
  }
  const int64_t e = finv[rowval[p0]] / M;
  for (int32_t p = p0; p < p1; ++p)
    if (finv[rowval[p]] / M != e) flags[1] = 1;  // an interior coarse node serves one element only
  // provisional slot: order of arrival; sorted into ascending column order by chain_coarse_sort_kernel
  const int32_t s = atomicAdd(&slot_counter[e], 1);
  if (s >= mc - 1) {
    flags[1] = 1;
    return;
  }
  cperm[e * mc + 1 + s] = (int32_t)c;
}

// ascending order of the interior coarse nodes inside every block, then the inverse map
__global__ __launch_bounds__(kSetupThreads) void chain_coarse_sort_kernel(int64_t nec, int mc, int32_t* __restrict__ cperm,
                                                                          int32_t* __restrict__ cinv,
                                                                          const int32_t* __restrict__ slot_counter,
                                                                          int64_t nel, int* __restrict__ flags) {
  const int64_t e = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (e >= nec) return;
  const int want = e < nel ? mc - 1 : 0;  // the trailing block holds the last vertex only
  if (slot_counter[e] != want || cperm[e * mc] < 0) {
    flags[1] = 1;
    return;
  }
  int32_t* v = cperm + e * mc + 1;
  for (int i = 1; i < want; ++i) {  // insertion sort, mc <= 8
    const int32_t key = v[i];
    int j = i - 1;
    while (j >= 0 && v[j] > key) {
      v[j + 1] = v[j];
      --j;
    }
    v[j + 1] = key;
  }
  for (int i = 0; i < want; ++i) cinv[v[i]] = (int32_t)(e * mc + 1 + i);
}

// chain transfer rows l[(q)][mc + 1]; bad[0] raised on a non-zero outside the pattern
__global__ __launch_bounds__(kSetupThreads) void chain_transfer_scatter_kernel(int64_t Nc, int M, int mc,
                                                                               const int32_t* __restrict__ finv,
                                                                               const int32_t* __restrict__ cinv,
                                                                               const int32_t* __restrict__ colptr,
                                                                               const int32_t* __restrict__ rowval,
                                                                               const double* __restrict__ vals,
                                                                               double* __restrict__ l, int* __restrict__ bad) {
  const int64_t c = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (c >= Nc) return;
  const int64_t qc = cinv[c];
  const int64_t ce = qc / mc;
  const int cj = (int)(qc - ce * mc);
  const int w = mc + 1;
  for (int32_t p = colptr[c]; p < colptr[c + 1]; ++p) {
    const int64_t q = finv[rowval[p]];
    const int64_t e = q / M;
    if (ce == e)
      l[q * w + cj] = vals[p];
    else if (ce == e + 1 && cj == 0)
      l[q * w + mc] = vals[p];
    else if (vals[p] != 0.0)
      bad[0] = 1;
  }
}

// agglomerating transfer of a chain level: l[(q)][mc], lp[e][mc]; bad[0] raised on a non-zero outside the pattern
__global__ __launch_bounds__(kSetupThreads) void agg_transfer_scatter_kernel(int64_t Nc, int M, int mc, int64_t rho,
                                                                             const int32_t* __restrict__ finv,
                                                                             const int32_t* __restrict__ colptr,
                                                                             const int32_t* __restrict__ rowval,
                                                                             const double* __restrict__ vals,
                                                                             double* __restrict__ l, double* __restrict__ lp,
                                                                             int* __restrict__ bad) {
  const int64_t c = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (c >= Nc) return;
  const int64_t Jc = c / mc;
  const int cj = (int)(c - Jc * mc);
  for (int32_t p = colptr[c]; p < colptr[c + 1]; ++p) {
    const int64_t q = finv[rowval[p]];
    const int64_t e = q / M;
    const int i = (int)(q - e * M);
    const int64_t J = e / rho;
    if (Jc == J)
      l[q * mc + cj] = vals[p];
    else if (i == 0 && Jc == J - 1 && e == J * rho)
      lp[e * mc + cj] = vals[p];
    else if (vals[p] != 0.0)
      bad[0] = 1;
  }
}

}  // namespace aggmg
