// Device kernels of libaggmg_hip (gfx950 / CDNA4, wave64, fp64).
//
// Two families:
//  * generic CSR row kernels (any assembled matrix: CG levels, unstructured transfers,
//    Schwarz blocks) -- LPR lanes per row, coalesced index/value reads, __shfl_xor reduction;
//  * the LDS-tiled fused block-tridiagonal ("BTD") kernel for DG / agglomerated-DG levels:
//    one workgroup owns a tile of elements plus a halo, keeps the iterate in LDS and the
//    operator rows in registers, and runs  [prolong-add] -> S block-Jacobi sweeps ->
//    [residual -> restriction]  on one pass over HBM (temporal blocking: the operator is read
//    once per launch, not once per sweep).
//
// The path is HBM-bound fp64 streaming (0.17 flop/B): no MFMA on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aggmg {

constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------
// generic CSR
// ------------------------------------------------------------------------------------------
struct CsrView {
  const int32_t* rowptr;
  const int32_t* colind;
  const double* vals;
  int64_t nrows;
};

enum CsrMode : int {
  kSpmvSet = 0,   // y = A x                      (restriction with the transposed orientation)
  kSpmvAdd = 1,   // y += A x                     (prolongation-add,     src/solvers.jl:42)
  kResidual = 2,  // y = b - A x                  (src/solvers.jl:33,36,44)
  kJacobi = 3     // y = x + alpha*((b - A x)/d)  (JacobiSmoother,       src/smoother.jl:56-58)
};

// LPR lanes cooperate on one row; rows are contiguous across the workgroup so rowptr / b / y
// accesses are coalesced and the value / index streams are read in LPR-wide contiguous pieces.
template <int LPR, int MODE>
__global__ __launch_bounds__(kThreads) void csr_row_kernel(CsrView A, const double* __restrict__ x,
                                                           const double* __restrict__ b,
                                                           const double* __restrict__ dg,
                                                           double alpha, double* __restrict__ y) {
  constexpr int kRows = kThreads / LPR;
  const int64_t row = (int64_t)blockIdx.x * kRows + threadIdx.x / LPR;
  const int sub = threadIdx.x % LPR;
  double acc = 0.0;
  if (row < A.nrows) {
    const int p0 = A.rowptr[row], p1 = A.rowptr[row + 1];
    for (int p = p0 + sub; p < p1; p += LPR) acc += A.vals[p] * x[A.colind[p]];
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, LPR);
  if (row < A.nrows && sub == 0) {
    if (MODE == kSpmvSet) y[row] = acc;
    if (MODE == kSpmvAdd) y[row] += acc;
    if (MODE == kResidual) y[row] = b[row] - acc;
    if (MODE == kJacobi) {
      const double r = b[row] - acc;
      const double yy = r / dg[row];
      y[row] = x[row] + alpha * yy;
    }
  }
}

// "CSR-stream" variant for short rows (a few to a few dozen entries per row: CG and DG stiffness
// matrices, transfers): a workgroup owns a run of consecutive rows holding at most kStreamNnz
// entries (row blocks cut on the host at upload), streams their values / column indices with
// fully coalesced loads, parks the products a_ij * x_j in LDS and lets one thread per row add its
// products in ascending column order -- the order SparseArrays' CSC scatter accumulates them.
// A row longer than kStreamNnz forms a block of its own and is reduced across the workgroup.
// Block shape (r04, held against rocSPARSE's adaptive csrmv on the same operators, tools/exp_rocsparse_calib.py: 69 us for
// y = A x on config 2's matrix): 4096 entries and up to 1024 rows per block -- 32 KB of LDS, five workgroups per CU, a
// reduction phase in which a thread walks up to four rows -- ran the residual at 105 us; at most ONE row per thread
// (256 rows) and 2048 entries 82 us, and with the entry streams loaded non-temporally (they are used once; the
// gathered x stays in the caches) 75 us = 0.70 of 8 TB/s, the vendor kernel's rate.  1024 entries: 92 us; 1536 / 1792 /
// 2048: 75 - 76 (1536 = 12 KB of LDS kept).  Non-temporal stores of y: no gain.
#ifndef AGGMG_STREAM_NNZ
#define AGGMG_STREAM_NNZ 1536
#endif
constexpr int kStreamNnz = AGGMG_STREAM_NNZ;
#ifndef AGGMG_STREAM_ROWS
#define AGGMG_STREAM_ROWS kThreads
#endif
constexpr int kStreamRows = AGGMG_STREAM_ROWS;   // most rows of a block
#ifndef AGGMG_STREAM_NTLOAD
#define AGGMG_STREAM_NTLOAD 1
#endif
#if AGGMG_STREAM_NTLOAD
#define AGGMG_SLD(p) __builtin_nontemporal_load(&(p))
#else
#define AGGMG_SLD(p) (p)
#endif
constexpr int kBandNnz = 2048;   // csr_band_kernel: most entries of a tile (block + halo rows); kBandNnz / kThreads per thread in registers
template <int MODE>
__global__ __launch_bounds__(kThreads) void csr_stream_kernel(CsrView A, const int32_t* __restrict__ rowblk,
                                                              const double* __restrict__ x,
                                                              const double* __restrict__ b,
                                                              const double* __restrict__ dg, double alpha,
                                                              double* __restrict__ y) {
  __shared__ double prod[kStreamNnz];
  __shared__ double red[kThreads / 64];
  const int tid = threadIdx.x;
  const int r0 = rowblk[blockIdx.x], r1 = rowblk[blockIdx.x + 1];
  const int p0 = A.rowptr[r0], p1 = A.rowptr[r1];
  const int nn = p1 - p0;
  auto finish = [&](int row, double acc) {
    if (MODE == kSpmvSet) y[row] = acc;
    if (MODE == kSpmvAdd) y[row] += acc;
    if (MODE == kResidual) y[row] = b[row] - acc;
    if (MODE == kJacobi) {
      const double r = b[row] - acc;
      const double yy = r / dg[row];
      y[row] = x[row] + alpha * yy;
    }
  };
  if (nn <= kStreamNnz) {
    // four independent (value, index, gather) chains in flight per thread
    int p = tid;
    for (; p + 3 * kThreads < nn; p += 4 * kThreads) {
      const int c0 = AGGMG_SLD(A.colind[p0 + p]), c1 = AGGMG_SLD(A.colind[p0 + p + kThreads]), c2 = AGGMG_SLD(A.colind[p0 + p + 2 * kThreads]),
                c3 = AGGMG_SLD(A.colind[p0 + p + 3 * kThreads]);
      const double v0 = AGGMG_SLD(A.vals[p0 + p]), v1 = AGGMG_SLD(A.vals[p0 + p + kThreads]), v2 = AGGMG_SLD(A.vals[p0 + p + 2 * kThreads]),
                   v3 = AGGMG_SLD(A.vals[p0 + p + 3 * kThreads]);
      prod[p] = v0 * x[c0];
      prod[p + kThreads] = v1 * x[c1];
      prod[p + 2 * kThreads] = v2 * x[c2];
      prod[p + 3 * kThreads] = v3 * x[c3];
    }
    for (; p < nn; p += kThreads) prod[p] = AGGMG_SLD(A.vals[p0 + p]) * x[AGGMG_SLD(A.colind[p0 + p])];
    __syncthreads();
    for (int r = r0 + tid; r < r1; r += kThreads) {
      const int q0 = A.rowptr[r] - p0, q1 = A.rowptr[r + 1] - p0;
      double acc = 0.0;
      for (int q = q0; q < q1; ++q) acc += prod[q];
      finish(r, acc);
    }
  } else {  // one long row
    double acc = 0.0;
    for (int p = p0 + tid; p < p1; p += kThreads) acc += A.vals[p] * x[A.colind[p]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < kThreads / 64; ++w) t += red[w];
      finish(r0, t);
    }
  }
}

// "CSR-row-thread" variant for matrices whose rows are ALL short (at most kRowThreadMax entries: DG / CG stiffness
// matrices, transfers): one thread per row walks its entries in ascending column order -- products rounded, then added,
// exactly the (product into LDS, sum from LDS) arithmetic of csr_stream_kernel and of SparseArrays' CSC scatter, so the
// two kernels agree bit for bit.  No LDS, no barrier, no row-block table: consecutive threads read consecutive short
// runs of the entry streams, every fetched line is used in full within a few iterations (from L1), four entries in
// flight per thread.
constexpr int kRowThreadMax = 32;
template <int MODE>
__global__ __launch_bounds__(kThreads) void csr_rowthread_kernel(CsrView A, const double* __restrict__ x,
                                                                 const double* __restrict__ b,
                                                                 const double* __restrict__ dg, double alpha,
                                                                 double* __restrict__ y) {
  const int64_t row = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (row >= A.nrows) return;
  const int p0 = A.rowptr[row], p1 = A.rowptr[row + 1];
  double acc = 0.0;
  {
#pragma clang fp contract(off)   // a product and its addition stay two roundings (the stream kernel's, the reference's)
    int p = p0;
    for (; p + 3 < p1; p += 4) {
      const int c0 = A.colind[p], c1 = A.colind[p + 1], c2 = A.colind[p + 2], c3 = A.colind[p + 3];
      const double v0 = A.vals[p], v1 = A.vals[p + 1], v2 = A.vals[p + 2], v3 = A.vals[p + 3];
      const double t0 = v0 * x[c0], t1 = v1 * x[c1], t2 = v2 * x[c2], t3 = v3 * x[c3];
      acc = acc + t0;
      acc = acc + t1;
      acc = acc + t2;
      acc = acc + t3;
    }
    for (; p < p1; ++p) {
      const double t = A.vals[p] * x[A.colind[p]];
      acc = acc + t;
    }
  }
  if (MODE == kSpmvSet) y[row] = acc;
  if (MODE == kSpmvAdd) y[row] += acc;
  if (MODE == kResidual) y[row] = b[row] - acc;
  if (MODE == kJacobi) {
    const double r = b[row] - acc;
    const double yy = r / dg[row];
    y[row] = x[row] + alpha * yy;
  }
}

// The same for square operators whose entries lie within bw <= kBandMaxBw of the diagonal (every DG / agglomerated
// operator in its own numbering): the window of x a workgroup's 256 rows reach is read ONCE, coalesced, into LDS and
// the per-entry gathers x[col] -- a third of the kernel's vector-memory instructions -- become LDS reads.  Same
// arithmetic, same bits.
template <int MODE>
__global__ __launch_bounds__(kThreads) void csr_rowthread_band_kernel(CsrView A, int bw, const double* __restrict__ x,
                                                                      const double* __restrict__ b,
                                                                      const double* __restrict__ dg, double alpha,
                                                                      double* __restrict__ y) {
  __shared__ double xw[kThreads + 2 * 32];   // kBandMaxBw = 32 (defined below)
  const int64_t r0 = (int64_t)blockIdx.x * kThreads;
  const int64_t w0 = r0 - bw > 0 ? r0 - bw : 0;
  const int64_t w1 = r0 + kThreads + bw < A.nrows ? r0 + kThreads + bw : A.nrows;
  for (int64_t w = w0 + threadIdx.x; w < w1; w += kThreads) xw[w - w0] = x[w];
  __syncthreads();
  const int64_t row = r0 + threadIdx.x;
  if (row >= A.nrows) return;
  const int p0 = A.rowptr[row], p1 = A.rowptr[row + 1];
  double acc = 0.0;
  {
#pragma clang fp contract(off)
    int p = p0;
    for (; p + 3 < p1; p += 4) {
      const int c0 = A.colind[p], c1 = A.colind[p + 1], c2 = A.colind[p + 2], c3 = A.colind[p + 3];
      const double v0 = A.vals[p], v1 = A.vals[p + 1], v2 = A.vals[p + 2], v3 = A.vals[p + 3];
      const double t0 = v0 * xw[c0 - w0], t1 = v1 * xw[c1 - w0], t2 = v2 * xw[c2 - w0], t3 = v3 * xw[c3 - w0];
      acc = acc + t0;
      acc = acc + t1;
      acc = acc + t2;
      acc = acc + t3;
    }
    for (; p < p1; ++p) {
      const double t = A.vals[p] * xw[A.colind[p] - w0];
      acc = acc + t;
    }
  }
  if (MODE == kSpmvSet) y[row] = acc;
  if (MODE == kSpmvAdd) y[row] += acc;
  if (MODE == kResidual) y[row] = b[row] - acc;
  if (MODE == kJacobi) {
    const double r = b[row] - acc;
    const double yy = r / dg[row];
    y[row] = xw[row - w0] + alpha * yy;
  }
}

// "CSR-band" variant for square operators whose entries all lie within `bw` of the diagonal (every DG / agglomerated
// operator of the reference in its own numbering; detected on the device at upload): S point-Jacobi sweeps per launch
// by temporal blocking, as the fused block-tridiagonal kernel does -- a workgroup owns a run of rows plus (S - 1) * bw
// rows of halo on either side, reads the window of x it needs ONCE, coalesced, into LDS, and after S sweeps the block's
// own rows are exact.  Products are summed per row in ascending column order (SparseArrays' CSC scatter order), so S
// sweeps in one launch give bit for bit what S launches give.  Row blocks (bandblk) are cut on the host (band_row_blocks).
constexpr int kBandSweeps = 8;   // most sweeps per launch (an operator's own limit, CsrDev::band_sweeps: 1 + 32 / bw, at least 2)
constexpr int kBandMaxBw = 32;   // widest band the window kernels take (csr_rowthread_band_kernel sizes its LDS for it)
static_assert(kBandMaxBw == 32, "csr_rowthread_band_kernel's window");
constexpr int kBandWin = kThreads + 2 * kBandMaxBw;   // a tile's rows (block + halo, <= kThreads) and bw columns beyond them
template <int MODE>
__global__ __launch_bounds__(kThreads) void csr_band_kernel(CsrView A, const int32_t* __restrict__ bandblk, int bw, int S,
                                                            const double* __restrict__ x, const double* __restrict__ b,
                                                            const double* __restrict__ dg, double alpha,
                                                            double* __restrict__ y, double* __restrict__ rout) {
  static_assert(MODE == kJacobi, "the multi-sweep window kernel runs point-Jacobi sweeps");
  // rout (may be null): the residual b - A y of the swept iterate on the block's rows, one more pass over the tile and
  // one more bw of halo -- the descent of a generic banded level (sweeps, then the residual for the restriction,
  // src/solvers.jl:32-36) in ONE pass over the operator; the same products and row sums as the stream kernel's residual
  // r04: a tile is the block's rows PLUS its (S - 1) * bw halo rows on either side, at most kThreads rows and kBandNnz
  // entries (cut at upload, band_row_blocks): one row per thread in every sweep, and the tile's entries -- values and
  // window-relative columns -- live in REGISTERS for the whole launch (kBandNnz / kThreads per thread, loaded once,
  // coalesced, non-temporally), as do the row's b, diagonal and entry range.  A sweep is then LDS work only: products
  // out of the x window, a barrier, one row sum per thread, the new value into the other window buffer, a barrier.  Every
  // row of the tile is updated in every sweep; what is wrong near the tile's ends (stale neighbours) moves inwards by bw
  // rows per sweep and stops short of the block's own rows, as in the fused tile kernels.  (Before: the entries were
  // re-read from memory in every sweep, 52 KB of LDS per workgroup, up to four rows per thread: 3 / 4 sweeps per launch
  // ran at 81 / 75 us per sweep on config 2's matrix, no better than single-sweep launches of the stream kernel.)
  constexpr int K = kBandNnz / kThreads;
  __shared__ double prod[kBandNnz];
  __shared__ double xw[2][kBandWin];
  const int tid = threadIdx.x;
  const int N = (int)A.nrows;
  const int r0 = bandblk[blockIdx.x], r1 = bandblk[blockIdx.x + 1];
  const int P = S + (rout ? 1 : 0);   // passes over the tile
  const int H = (P - 1) * bw;
  const int ra = max(0, r0 - H), rb = min(N, r1 + H);
  const int w0 = max(0, ra - bw), w1 = min(N, rb + bw);
  const int pa = A.rowptr[ra], nn = A.rowptr[rb] - pa;
  double v[K];
  int c[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int p = tid + k * kThreads;
    v[k] = 0.0;
    c[k] = 0;
    if (p < nn) {
      v[k] = __builtin_nontemporal_load(&A.vals[pa + p]);
      c[k] = __builtin_nontemporal_load(&A.colind[pa + p]) - w0;
    }
  }
  const int r = ra + tid;
  const bool has = r < rb;
  int q0 = 0, q1 = 0;
  double br = 0.0, dr = 1.0;
  if (has) {
    q0 = A.rowptr[r] - pa;
    q1 = A.rowptr[r + 1] - pa;
    br = b[r];
    dr = dg[r];
  }
  // (the window cells beyond the tile's rows are never written again: both buffers hold them)
  for (int w = w0 + tid; w < w1; w += kThreads) {
    const double xv = x[w];
    xw[0][w - w0] = xv;
    xw[1][w - w0] = xv;
  }
  __syncthreads();
  int cur = 0;
  for (int s = 0; s < P; ++s) {
    const double* xa = xw[cur];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int p = tid + k * kThreads;
      if (p < nn) prod[p] = v[k] * xa[c[k]];
    }
    __syncthreads();
    if (has) {
      const bool own = r >= r0 && r < r1;
      double acc = 0.0;
      for (int q = q0; q < q1; ++q) acc += prod[q];
      const double res = br - acc;
      if (s < S) {
        const double yy = res / dr;
        const double val = xa[r - w0] + alpha * yy;
        if (s == S - 1 && own) y[r] = val;
        if (s < P - 1) xw[cur ^ 1][r - w0] = val;
      } else if (own) {
        rout[r] = res;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

// largest |row - col| over the stored entries of a square CSR (atomicMax into *out)
static __global__ __launch_bounds__(kThreads) void csr_bandwidth_kernel(CsrView A, int* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  int m = 0;
  if (row < A.nrows) {
    const int p0 = A.rowptr[row], p1 = A.rowptr[row + 1];
    if (p1 > p0) m = max((int)row - A.colind[p0], A.colind[p1 - 1] - (int)row);   // columns ascend within a row
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

// y[inds] (+)= Binv_blk * r[inds]   -- generic block apply (arbitrary, possibly overlapping,
// index lists: BlockJacobi / AdditiveSchwarz / HybridSchwarz, src/smoother.jl:6-18,30-46,69-81).
// One thread per (block, local row).  ATOMIC for overlapping blocks (y pre-zeroed).
template <bool ATOMIC>
__global__ __launch_bounds__(kThreads) void block_apply_kernel(const double* __restrict__ binv,
                                                               const int32_t* __restrict__ inds,
                                                               int m, int64_t nb,
                                                               const double* __restrict__ r,
                                                               double* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t blk = t / m;
  const int i = (int)(t - blk * m);
  if (blk >= nb) return;
  const double* Bi = binv + (blk * m + i) * m;
  const int32_t* id = inds + blk * m;
  double acc = 0.0;
  for (int j = 0; j < m; ++j) acc += Bi[j] * r[id[j]];
  if (ATOMIC)
    atomicAdd(&y[id[i]], acc);
  else
    y[id[i]] = acc;
}

// One damped sweep of a block smoother on arbitrary index lists, u <- u + alpha * sum_k scatter(B_k^{-1} (b - A u)[inds_k])
// (apply_smoother of BlockJacobi / AdditiveSchwarzSmoother / HybridSchwarzSmoother inside the sweep loop,
// src/smoother.jl:6-46,69-81, src/solvers.jl:32-34), without the residual vector ever reaching HBM: a workgroup takes
// kThreads / m whole blocks, thread (block, i) forms residual row inds[block][i] straight from the CSR (its entries in
// ascending column order, product and sum rounded apart: csr_rowthread_kernel's arithmetic), the block's residual goes
// through LDS, and the thread applies row i of the block inverse.
//   DIRECT: the lists partition the rows (every row in exactly one block): out[row] = u[row] + alpha * y, one launch;
//           out must not alias u (rows of other blocks are still being read).
//   else:   Y[block * m + i] = y, and block_combine_kernel adds up the entries covering each row -- in list order, no
//           atomics, no zeroing: the same bits run to run.
// (Non-temporal loads of the entry streams, which lifted the stream kernel, cost this one 36 % -- 283 against 208 us per
// Schwarz sweep: a thread walks its row entry by entry and lives on the lines staying in the cache between its iterations.)
template <bool DIRECT>
__global__ __launch_bounds__(kThreads) void block_sweep_kernel(CsrView A, const double* __restrict__ binv,
                                                               const int32_t* __restrict__ inds, int m, int64_t nb,
                                                               const double* __restrict__ u, const double* __restrict__ b,
                                                               double alpha, double* __restrict__ out) {
  __shared__ double rl[kThreads];
  const int bpw = kThreads / m;            // blocks per workgroup
  const int tid = threadIdx.x;
  const int lb = tid / m, i = tid - lb * m;
  const int64_t blk = (int64_t)blockIdx.x * bpw + lb;
  const bool act = lb < bpw && blk < nb;
  int32_t id = 0;
  if (act) {
    id = inds[blk * m + i];
    const int p0 = A.rowptr[id], p1 = A.rowptr[id + 1];
    double acc = 0.0;
    {
#pragma clang fp contract(off)
      int p = p0;
      for (; p + 3 < p1; p += 4) {
        const int c0 = A.colind[p], c1 = A.colind[p + 1], c2 = A.colind[p + 2], c3 = A.colind[p + 3];
        const double v0 = A.vals[p], v1 = A.vals[p + 1], v2 = A.vals[p + 2], v3 = A.vals[p + 3];
        const double t0 = v0 * u[c0], t1 = v1 * u[c1], t2 = v2 * u[c2], t3 = v3 * u[c3];
        acc = acc + t0;
        acc = acc + t1;
        acc = acc + t2;
        acc = acc + t3;
      }
      for (; p < p1; ++p) {
        const double t = A.vals[p] * u[A.colind[p]];
        acc = acc + t;
      }
    }
    rl[tid] = b[id] - acc;
  }
  __syncthreads();
  if (!act) return;
  const double* Bi = binv + (blk * m + i) * m;
  const double* rb = rl + lb * m;
  double y = 0.0;
  for (int j = 0; j < m; ++j) y += Bi[j] * rb[j];
  if (DIRECT) {
    const double v = alpha * y;
    out[id] = u[id] + v;
  } else {
    out[blk * m + i] = y;
  }
}

// out[row] = u[row] + alpha * (sum of the Y entries covering the row) / (cnt ? cnt[row] : 1); the covering entries of a
// row (cover_ptr / cover_idx: flat indices block * m + i, ascending) come from the smoother's index lists, rows in no
// block keep their value (u null: zero).  out may alias u.
static __global__ __launch_bounds__(kThreads) void block_combine_kernel(int64_t n, const int32_t* __restrict__ cover_ptr,
                                                                        const uint32_t* __restrict__ cover_idx,
                                                                        const double* __restrict__ Y, const double* u,
                                                                        const double* __restrict__ cnt, double alpha, double* out) {
  const int64_t row = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (row >= n) return;
  double v = 0.0;
  for (int q = cover_ptr[row]; q < cover_ptr[row + 1]; ++q) v += Y[cover_idx[q]];
  if (cnt) v = v / cnt[row];
  v = alpha * v;
  out[row] = u ? u[row] + v : v;
}

// Y[block * m + i] = (Binv_block r[inds_block])_i : the block results of overlapping lists kept apart, for
// block_combine_kernel to add up in list order (apply_smoother of the Schwarz smoothers without atomics)
static __global__ __launch_bounds__(kThreads) void block_apply_flat_kernel(const double* __restrict__ binv,
                                                                           const int32_t* __restrict__ inds, int m, int64_t nb,
                                                                           const double* __restrict__ r, double* __restrict__ Y) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t blk = t / m;
  const int i = (int)(t - blk * m);
  if (blk >= nb) return;
  const double* Bi = binv + (blk * m + i) * m;
  const int32_t* id = inds + blk * m;
  double acc = 0.0;
  for (int j = 0; j < m; ++j) acc += Bi[j] * r[id[j]];
  Y[t] = acc;
}

// out = (u ? u : 0) + alpha * (y / (cnt ? cnt : 1)); out may alias u (block smoothers update in place)
static __global__ __launch_bounds__(kThreads) void axpy_scaled_kernel(int64_t n, const double* u,
                                                                      const double* __restrict__ y,
                                                                      const double* __restrict__ cnt,
                                                                      double alpha, double* out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  double v = y[i];
  if (cnt) v = v / cnt[i];
  v = alpha * v;
  out[i] = u ? u[i] + v : v;
}

// ------------------------------------------------------------------------------------------
// up to four strided 2-D copies in one launch (interface packing / unpacking around the
// collectives of element-partitioned runs: one launch instead of one per slice)
// ------------------------------------------------------------------------------------------
struct CopySegs {
  const double* src[4];
  double* dst[4];
  int64_t rows[4], cols[4], src_ld[4], dst_ld[4];
};

static __global__ __launch_bounds__(kThreads) void copy_segments_kernel(CopySegs S) {
  const int g = blockIdx.y;
  const int64_t n = S.rows[g] * S.cols[g];
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += (int64_t)gridDim.x * kThreads) {
    const int64_t i = t / S.cols[g], j = t - i * S.cols[g];
    S.dst[g][i * S.dst_ld[g] + j] = S.src[g][i * S.src_ld[g] + j];
  }
}

// ------------------------------------------------------------------------------------------
// outer-solver vector kernels (multigrid / iterative_smoother_solve loops, PCG): the scalars stay
// on the device, the host only reads the residual norm it needs for the stopping test
// ------------------------------------------------------------------------------------------
constexpr int kDotBlocks = 1024;  // fixed grid and fixed summation tree: results are reproducible

// partial[b] = sum_{i in slice b} x_i y_i   (slice = contiguous n / gridDim range)
static __global__ __launch_bounds__(kThreads) void dot_partial_kernel(int64_t n, const double* __restrict__ x,
                                                               const double* __restrict__ y,
                                                               double* __restrict__ partial) {
  __shared__ double sh[kThreads];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < n ? lo + per : n;
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) acc += x[i] * y[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// partial[b] = sum_{i in slice b} (x_i - y_i)^2   -- ||x - u_exact||_2 of the `err` histories (src/solvers.jl:128,202)
static __global__ __launch_bounds__(kThreads) void diff2_partial_kernel(int64_t n, const double* __restrict__ x,
                                                                 const double* __restrict__ y,
                                                                 double* __restrict__ partial) {
  __shared__ double sh[kThreads];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < n ? lo + per : n;
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
    const double d = x[i] - y[i];
    acc += d * d;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// out[0] = sum of the partials (one workgroup), optionally its square root
static __global__ __launch_bounds__(kThreads) void dot_final_kernel(int nparts, const double* __restrict__ partial,
                                                             double* __restrict__ out, int take_sqrt) {
  __shared__ double sh[kThreads];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kThreads) acc += partial[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = take_sqrt ? sqrt(sh[0]) : sh[0];
}

// The tile sums of a checkpoint launch -> out[k] = (sqrt(sum_t part[k][t][0]), sqrt(sum_t part[k][t][1])) per checkpoint k,
// in two stages with a fixed summation order (hundreds of thousands of tiles at 2^24 elements: one workgroup walking
// them alone took 0.5 - 3 ms, latency-bound): stage 1, grid (G, checkpoints): workgroup g sums tiles g * 256 + tid,
// + G * 256, ... into mid[k][g][2]; stage 2, one workgroup per checkpoint: the G sums, root.
static __global__ __launch_bounds__(kThreads) void chk_reduce1_kernel(int64_t ntiles, const double* __restrict__ part,
                                                                    double* __restrict__ mid) {
  __shared__ double sh[2][kThreads];
  part += (int64_t)blockIdx.y * ntiles * 2;
  double a0 = 0.0, a1 = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < ntiles; t += (int64_t)gridDim.x * kThreads) {
    a0 += part[2 * t];
    a1 += part[2 * t + 1];
  }
  sh[0][threadIdx.x] = a0;
  sh[1][threadIdx.x] = a1;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* o = mid + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
    o[0] = sh[0][0];
    o[1] = sh[1][0];
  }
}
static __global__ __launch_bounds__(kThreads) void chk_reduce2_kernel(int G, const double* __restrict__ mid, double* __restrict__ out) {
  __shared__ double sh[2][kThreads];
  mid += (int64_t)blockIdx.x * G * 2;
  double a0 = 0.0, a1 = 0.0;
  for (int t = threadIdx.x; t < G; t += kThreads) {
    a0 += mid[2 * t];
    a1 += mid[2 * t + 1];
  }
  sh[0][threadIdx.x] = a0;
  sh[1][threadIdx.x] = a1;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = sqrt(sh[0][0]);
    out[blockIdx.x * 2 + 1] = sqrt(sh[1][0]);
  }
}
constexpr int kChkReduceGroups = 256;   // stage-1 workgroups per checkpoint at most

// PCG step with q = -A p (the residual kernel's sign):  a = rz / (-(p.q));  x += a p;  r += a q
static __global__ __launch_bounds__(kThreads) void pcg_xr_kernel(int64_t n, double* __restrict__ x, double* __restrict__ r,
                                                          const double* __restrict__ p,
                                                          const double* __restrict__ q,
                                                          const double* __restrict__ rz,
                                                          const double* __restrict__ pq) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const double a = rz[0] / (-pq[0]);
  x[i] += a * p[i];
  r[i] += a * q[i];
}

// p = z + (rz_new / rz_old) p
static __global__ __launch_bounds__(kThreads) void pcg_p_kernel(int64_t n, double* __restrict__ p,
                                                         const double* __restrict__ z,
                                                         const double* __restrict__ rz_new,
                                                         const double* __restrict__ rz_old) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  p[i] = z[i] + (rz_new[0] / rz_old[0]) * p[i];
}

// ------------------------------------------------------------------------------------------
// fused block-tridiagonal kernel
// ------------------------------------------------------------------------------------------
// Level data, all fp64, row-major per DoF row (row = e*M + i):
//   binv [N][M]  rows of the inverse of the diagonal block  B_e^{-1}
//   dblk [N][M]  rows of the diagonal block                 D_e  (= B_e)
// CMP (nodal DG): the sub-diagonal block has ONE non-zero column c_sub, the super-diagonal
// block ONE non-zero row r_sup:
//   scol [N]     Sub_e[:, c_sub]        pcol [N]   (B_e^{-1} Sub_e)[:, c_sub]
//   qrow [ne][M] Sup_e[r_sup, :]
// dense (agglomerated levels, any pattern):
//   sub, sup [N][M] rows of Sub_e, Sup_e;   P, Q [N][M] rows of B_e^{-1}Sub_e, B_e^{-1}Sup_e
// With these, one damped block-Jacobi sweep  u <- u + alpha*B^{-1}(b - A u)  is the stencil
//   u_e <- u_e + alpha*( g_e - P_e u_{e-1} - Q_e u_{e+1} - u_e ),   g_e = B_e^{-1} b_e
// (B^{-1}D = I is used algebraically; differs from the reference's LU solve at round-off only).
struct BtdLevel {
  const double *binv, *dblk;
  // symmetric operators (A = A' to round-off, detected at set-up): bsym [ne][M(M+1)/2] holds the
  // upper triangle of the symmetric B_e^{-1} instead of binv, and pcol is not stored at all: the
  // sub-diagonal column of element e is the super-diagonal row of element e-1
  // (scol_e = qrow_{e-1}), so pcol_e = B_e^{-1} qrow_{e-1} is formed in the kernel
  const double* bsym;
  const double *scol, *pcol, *qrow;
  const double *sub, *sup, *P, *Q;
  int64_t ne;
  int c_sub, r_sup;
};

struct FusedArgs {
  BtdLevel lv;
  const double* u_in;  // nullptr: iterate starts at zero (src/solvers.jl:29-31)
  const double* b;
  double* u_out;       // nullptr: iterate not stored (residual-only call)
  double alpha;
  int nsweeps;
  // optional prolongation-add before the sweeps: u += LF_in * uc   (src/solvers.jl:42)
  const double* lf_in;  // [N][mc_in] rows of L for the owning coarse element
  const double* lf1_in;  // [N] second entries of those rows when mc_in == 2 and every first entry is exactly 1.0
                         // (detected at set-up: the constant mode of an agglomerate's modal basis evaluated at the
                         // fine nodes, src/agglomerated_dg_mesh.jl:297-315) -- half the transfer bytes; else null
  const double* uc;
  int mc_in, rho_in;
  // optional residual after the sweeps (src/solvers.jl:36), stored and/or restricted
  int do_residual;
  double* r_out;         // may be nullptr
  const double* lf_out;  // [N][mc_out] rows of L: rc = L' r from the explicit residual
  const double* lf1_out; // [N] as lf1_in, for lf_out
  const double* ld_out;  // [N][mc_out] rows of (L_e' D_e)': rc = (L'D) w, w = B^{-1} r -- no D, no L read
  double* rc_out;        // restricted residual
  int mc_out, rho_out;
  // agglomerates of different sizes (rho_in / rho_out unused then): coarse element of every fine element,
  // first fine element of every coarse element.  agg_shift >= 0: the tiles' owned ranges are moved to agglomerate
  // boundaries -- the agglomerate holding a tile's nominal first element belongs to that tile as a whole, which costs
  // agg_shift = (largest agglomerate - 1) more elements of halo on the left -- so every agglomerate is restricted by
  // ONE tile with a plain store (no zeroing of rc_out, no atomics, run-to-run identical).  agg_shift < 0 (very large
  // agglomerates): an agglomerate cut by a tile boundary is restricted by both tiles, each adding its part atomically
  // (rc_out zeroed by the caller; two parts: order-independent).
  const int32_t* par_in;
  const int32_t* par_out;
  const int32_t* first_out;
  int64_t nec_out;  // coarse elements behind rc_out
  int agg_shift;
  // tiling
  int owned;      // owned elements per tile (multiple of rho_out)
  int halo_left;  // elements of halo on the left of the owned range
  // tile of workgroup b = b + (b >= tile_split ? tile_skip : 0): a launch may cover the tiles at
  // the two ends of the level, or the ones in between (element-partitioned runs produce the
  // interface elements first and exchange them under the interior launch)
  int tile_split;
  int64_t tile_skip;
  // 0: block-Jacobi sweeps.  1 / 2: red-black block Gauss-Seidel sweeps (EXTENSION, no reference
  // counterpart): even elements then odd ones (1), or odd then even (2); each half-sweep uses the
  // other colour's newest values and costs one element of halo.
  int gs;
  // Checkpoint (btd_fused_kernel<..., CHK = true> only; the loop of multigrid, src/solvers.jl:124-131, with its
  // residual test after EVERY cycle): after chk_sweep of the nsweeps sweeps -- between the post-smoothing of one cycle
  // and the pre-smoothing of the next, which share this launch -- the tile's sums of squares of b - A u and (chk_exact
  // not null) of u - u_exact over its owned rows go to chk_part[tile][2]: ||A x - b|| (:127) and ||x - u_exact|| (:128)
  // without a residual launch of their own.  The iterate itself is not stored: when the loop stops at this check, the
  // launch's input is still there and the ascent is run once more on its own.
  int chk_sweep;
  // more than one checkpoint per launch (iterative_smoother_solve's test after every sweep, src/solvers.jl:198-206):
  // after chk_sweep, chk_sweep + chk_stride, ... sweeps, and (chk_final) after the last one; checkpoint k's tile sums go
  // to chk_part[(k * chk_tiles + tile)][2] (chk_tiles: the launch's tile count, set by the launcher)
  int chk_stride;
  int chk_final;
  int64_t chk_tiles;
  const double* chk_exact;
  double* chk_part;
};

__device__ __forceinline__ int64_t fused_tile(const FusedArgs& a) {
  return (int64_t)blockIdx.x + ((int)blockIdx.x >= a.tile_split ? a.tile_skip : 0);
}

// streaming (read-once) operator data: optionally non-temporal so it does not displace the
// re-read halo / coarse vectors in L2
// AGGMG_NT bit 0: non-temporal loads of the read-once operator streams; bit 1: non-temporal
// stores of the iterate (tuning knobs, see tools/exp_fused.py)
#ifndef AGGMG_NT
#define AGGMG_NT 2  // measured: non-temporal iterate stores -2.6 % per V-cycle, NT operator loads no gain (r01); r04, on the
                    // symmetric-packed path -- packed inverses, closing residual's rows, b --: 19 % SLOWER (fine descent 1.15 ->
                    // 1.37 ms: the four lanes of an element read overlapping entries of the packed triangle in successive loads)
#endif
#if AGGMG_NT & 1
#define AGGMG_LD(p) __builtin_nontemporal_load(&(p))
#else
#define AGGMG_LD(p) (p)
#endif
#if AGGMG_NT & 2
#define AGGMG_ST(p, v) __builtin_nontemporal_store((v), &(p))
#else
#define AGGMG_ST(p, v) ((p) = (v))
#endif

// sum / broadcast inside the group of W consecutive lanes holding one element's rows (W = 2^k), no LDS traffic, no
// barrier.  W = 2, 4: the group lies inside a quad of lanes, so the exchange is a DPP quad permutation -- a VALU move
// modifier -- instead of ds_bpermute, which goes through the LDS pipe and costs its latency twice per sweep in the
// dependent chain (AGGMG_DPP=0 at compile time: the permute form, for A/B runs).  Same values, same order of additions.
#ifndef AGGMG_DPP
#define AGGMG_DPP 1
#endif
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int W>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (AGGMG_DPP && (W == 2 || W == 4)) {
    if constexpr (W == 4) v += quad_perm<0x4E>(v);   // lanes [2, 3, 0, 1]: xor 2
    v += quad_perm<0xB1>(v);                         // lanes [1, 0, 3, 2]: xor 1
    return v;
  } else {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, W);
    return v;
  }
}
template <int W>
__device__ __forceinline__ double group_bcast(double v, int j) {
  if constexpr (AGGMG_DPP && W == 4) {
    switch (j) {   // (j is a constant after unrolling: one quad permutation [j, j, j, j])
      case 0: return quad_perm<0x00>(v);
      case 1: return quad_perm<0x55>(v);
      case 2: return quad_perm<0xAA>(v);
      default: return quad_perm<0xFF>(v);
    }
  } else if constexpr (AGGMG_DPP && W == 2) {
    return j == 0 ? quad_perm<0xA0>(v) : quad_perm<0xF5>(v);   // [0, 0, 2, 2] / [1, 1, 3, 3]
  } else {
    return __shfl(v, j, W);
  }
}

// GS: red-black block Gauss-Seidel sweeps (FusedArgs::gs gives the colour order) instead of
// block-Jacobi ones -- a compile-time variant, so the block-Jacobi kernel carries none of it
// (AGGMG_CHK_WAVES: minimum waves per SIMD asked of the checkpoint variant; measured 7 and 6 -- spills into the sweep
// loop, 2.8 / 2.3 ms against 1.45 ms per launch -- so 1: no constraint)
#ifndef AGGMG_CHK_WAVES
#define AGGMG_CHK_WAVES 1
#endif
template <int M, bool CMP, int NS, bool SYM = false, int NT = kThreads, bool GS = false, bool CHK = false>
__global__ __launch_bounds__(NT, CHK ? AGGMG_CHK_WAVES : 1) void btd_fused_kernel(FusedArgs a) {
  static_assert(!SYM || M == 2 || M == 4 || M == 8, "symmetric packing needs the lane-group path");
  static_assert(!(CHK && GS), "the checkpoint is for block-Jacobi launches");
  // GRP: the rows of one element sit in M = 2^k adjacent lanes, so element-wide sums and
  // broadcasts (q.u+, B^{-1} b) go through cross-lane moves instead of LDS round trips
  constexpr bool GRP = CMP && (M == 2 || M == 4 || M == 8);
  // dense off-diagonal blocks of a symmetric operator: only the super-diagonal blocks are read
  // (Sub_e = Sup_{e-1}'), P = B^{-1}Sub and Q = B^{-1}Sup are formed in registers at load time
  constexpr bool DSYM = !CMP && SYM;
  constexpr int EPS = NT / M;  // elements per slab
  constexpr int TE = EPS * NS;       // elements per tile (owned + halos)
  extern __shared__ double lds[];
  // two iterate buffers, each padded by one zero element on both sides: index (x*M + j),
  // x in [-1, TE]
  double* buf0 = lds + M;
  double* buf1 = lds + (TE + 2) * M + M;

  const int tid = threadIdx.x;
  const bool active = tid < EPS * M;
  const int le = tid / M;
  const int i = tid - le * M;
  const int64_t ne = a.lv.ne;
  const int64_t e0 = fused_tile(a) * a.owned - a.halo_left;  // element at x = 0
  // owned elements x in [own0, own1) of the tile; with agglomerates of different sizes below, on agglomerate
  // boundaries (FusedArgs::agg_shift; two dependent index loads, issued here with the tile's other streams)
  int own0 = a.halo_left, own1 = a.halo_left + a.owned;
  if (a.par_out && a.agg_shift >= 0 && (a.lf_out || a.ld_out)) {
    const int64_t n0 = e0 + a.halo_left, n1 = n0 + a.owned;
    const int64_t s0 = n0 < ne ? a.first_out[a.par_out[n0]] : ne;
    const int64_t s1 = n1 < ne ? a.first_out[a.par_out[n1]] : ne;
    own0 = (int)(s0 - e0);
    own1 = (int)(s1 - e0);
  }

  if (tid < M) {
    buf0[-M + tid] = 0.0;
    buf0[TE * M + tid] = 0.0;
    buf1[-M + tid] = 0.0;
    buf1[TE * M + tid] = 0.0;
  }

  // the stencil form (B^{-1} rows, P, Q) is only needed when something is swept or restricted
  const bool need_g = a.nsweeps > 0 || a.ld_out != nullptr;
  // rows of L' (or (L'D)') for the restriction, fetched with the tile's other streams when the
  // coarse space has two modes per element (one 16-byte load per row)
  const double* lfo_pre = a.ld_out ? a.ld_out : a.lf_out;
  const bool pre2 = a.do_residual && lfo_pre && a.mc_out == 2;
  double l2x[NS], l2y[NS];
  double g[NS], bb[NS], uu[NS];
  double bi[NS][M];                              // B^{-1} rows, dead after g is formed
  double binv_r[NS], pc[NS], qv[NS][GRP ? 1 : M];  // CMP (GRP: own entry of the q row only)
  double Pr[NS][M], Qr[NS][M];                   // dense
  bool valid[NS];

  // ---- load phase: everything this tile needs from HBM, issued up front --------------------
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int x = s * EPS + le;
    const int64_t e = e0 + x;
    valid[s] = active && e >= 0 && e < ne;
    const int64_t row = e * M + i;
    uu[s] = 0.0;
    bb[s] = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) bi[s][j] = 0.0;
    if (valid[s]) {
      if (need_g) {
        if (SYM) {  // row i of the symmetric inverse out of its packed upper triangle
          constexpr int T = M * (M + 1) / 2;
#pragma unroll
          for (int j = 0; j < M; ++j) {
            const int lo_ = i < j ? i : j, hi_ = i < j ? j : i;
            bi[s][j] = a.lv.bsym[e * T + lo_ * M - (lo_ * (lo_ - 1)) / 2 + (hi_ - lo_)];
          }
        } else {
#pragma unroll
          for (int j = 0; j < M; ++j) bi[s][j] = AGGMG_LD(a.lv.binv[row * M + j]);
        }
      }
      bb[s] = a.b[row];
      if (a.u_in) uu[s] = a.u_in[row];
      if (pre2) {
        if (a.lf1_out && !a.ld_out) {   // unit first column: 1.0 * r is r, bit for bit what the stored 1.0 gives
          l2x[s] = 1.0;
          l2y[s] = AGGMG_LD(a.lf1_out[row]);
        } else {
          const double2 t2 = *reinterpret_cast<const double2*>(lfo_pre + row * 2);
          l2x[s] = t2.x;
          l2y[s] = t2.y;
        }
      }
      if (CMP) {
        if (SYM) {
          pc[s] = (need_g && e > 0) ? a.lv.qrow[(e - 1) * M + i] : 0.0;  // q_{e-1}[i]; B^{-1} applied below
        } else {
          pc[s] = need_g ? AGGMG_LD(a.lv.pcol[row]) : 0.0;
        }
        if (GRP) {
          qv[s][0] = AGGMG_LD(a.lv.qrow[e * M + i]);
        } else {
#pragma unroll
          for (int j = 0; j < (GRP ? 1 : M); ++j) qv[s][j] = a.lv.qrow[e * M + j];
        }
      } else if (DSYM) {
        // park Sup_{e-1}[i][:] in Pr and Sup_e[i][:] in Qr; turned into P, Q rows after the loop
#pragma unroll
        for (int j = 0; j < M; ++j) {
          Pr[s][j] = (need_g && e > 0) ? a.lv.sup[(row - M) * M + j] : 0.0;
          Qr[s][j] = need_g ? a.lv.sup[row * M + j] : 0.0;
        }
      } else {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          Pr[s][j] = need_g ? AGGMG_LD(a.lv.P[row * M + j]) : 0.0;
          Qr[s][j] = need_g ? AGGMG_LD(a.lv.Q[row * M + j]) : 0.0;
        }
      }
      if (a.lf_in) {  // u += L uc : J = e / rho, ascending mode order (CSC scatter order)
        const int64_t J = a.par_in ? (int64_t)a.par_in[e] : e / a.rho_in;
        double add = 0.0;
        if (a.mc_in == 2) {  // one 16-byte load each for the L row and the coarse pair
          typedef double v2d __attribute__((ext_vector_type(2)));
          double2 l2;
          if (a.lf1_in) {
            l2.x = 1.0;
            l2.y = AGGMG_LD(a.lf1_in[row]);
          } else {
            const v2d lv2 = AGGMG_LD(*reinterpret_cast<const v2d*>(a.lf_in + row * 2));
            l2.x = lv2.x;
            l2.y = lv2.y;
          }
          const double2 u2 = *reinterpret_cast<const double2*>(a.uc + J * 2);
          add = l2.x * u2.x;
          add += l2.y * u2.y;
        } else {
          for (int c = 0; c < a.mc_in; ++c) add += a.lf_in[row * a.mc_in + c] * a.uc[J * a.mc_in + c];
        }
        uu[s] += add;
      }
    } else {
      if (CMP) {
        pc[s] = 0.0;
#pragma unroll
        for (int j = 0; j < (GRP ? 1 : M); ++j) qv[s][j] = 0.0;
      } else {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          Pr[s][j] = 0.0;
          Qr[s][j] = 0.0;
        }
      }
    }
    binv_r[s] = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (j == a.lv.r_sup) binv_r[s] = bi[s][j];
    if (active) {
      buf0[x * M + i] = uu[s];
      if (!GRP && !DSYM) buf1[x * M + i] = bb[s];
    }
  }
  if (GRP || DSYM) {
    // g = B^{-1} b with the element's b_e broadcast across its lane group
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < M; ++j) acc += bi[s][j] * group_bcast<M>(bb[s], j);
      g[s] = acc;
      if (DSYM) {
        // P_i[j] = sum_k B^{-1}_ik Sup_{e-1}[j][k]   (lane j holds row j of Sup_{e-1})
        // Q_i[j] = sum_k B^{-1}_ik Sup_e[k][j]       (lane k holds row k of Sup_e)
        double pn[M], qn[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
          double pa = 0.0, qa = 0.0;
#pragma unroll
          for (int k = 0; k < M; ++k) {
            pa += bi[s][k] * group_bcast<M>(Pr[s][k], j);
            qa += bi[s][k] * group_bcast<M>(Qr[s][j], k);
          }
          pn[j] = pa;
          qn[j] = qa;
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
          Pr[s][j] = pn[j];
          Qr[s][j] = qn[j];
        }
      }
      if (SYM && CMP) {  // pcol = B^{-1} q_{e-1}
        const double qp = pc[s];
        double pacc = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j) pacc += bi[s][j] * group_bcast<M>(qp, j);
        pc[s] = pacc;
      }
    }
    __syncthreads();
  } else {
    __syncthreads();
    // g = B^{-1} b  (the element's whole b_e is read back from LDS)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      g[s] = 0.0;
      if (active) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j) acc += bi[s][j] * buf1[x * M + j];
        g[s] = acc;
      }
    }
    __syncthreads();
  }

  // ---- sweeps --------------------------------------------------------------------------------
  // block-Jacobi: LDS ping-pong.  Red-black block Gauss-Seidel (GS): in place, one colour per
  // half-sweep -- (B^{-1}(b - A u))_e = g_e - P u_{e-1} - Q u_{e+1} - u_e needs the row's own value
  // and the two neighbouring elements, which have the other colour and are not written in this
  // half-sweep; every thread evaluates the update, so the lane-group reductions stay convergent,
  // and only the current colour commits it.
  double* cur = buf0;
  double* nxt = buf1;
  const int nhalf = GS ? 2 * a.nsweeps : a.nsweeps;
  // CHK: the checkpoint of FusedArgs -- explicit residual of the iterate in `it` (the operator's own entries, ascending
  // column order: the final residual's expressions) on the owned rows, its square and that of u - u_exact summed over the
  // tile in a fixed order (wave by wave, then the four waves)
  [[maybe_unused]] int kchk = 0;
  // (CHK, CMP) the residual rows' operator entries -- the row of the diagonal block, the sub-diagonal column entry -- are
  // read from memory by the FIRST checkpoint only and parked in thread-private LDS slots behind the reduction scratch:
  // later checkpoints and the launch's closing residual (the restriction's) take them from there instead of a second pass
  // over the diagonal blocks (2.1 GB at 2^24 elements p = 3)
  [[maybe_unused]] double* const stash = lds + 2 * (TE + 2) * M + 2 * (NT / 64);
  [[maybe_unused]] bool stashed = false;
  [[maybe_unused]] auto row_entries = [&](int s, int64_t row, double (&dk)[M], double& sc) {
    if (CHK && CMP && stashed) {
#pragma unroll
      for (int j = 0; j < M; ++j) dk[j] = stash[(s * (M + 1) + j) * NT + tid];
      sc = stash[(s * (M + 1) + M) * NT + tid];
    } else {
      sc = a.lv.scol[row];
#pragma unroll
      for (int j = 0; j < M; ++j) dk[j] = a.lv.dblk[row * M + j];
      if constexpr (CHK && CMP) {
#pragma unroll
        for (int j = 0; j < M; ++j) stash[(s * (M + 1) + j) * NT + tid] = dk[j];
        stash[(s * (M + 1) + M) * NT + tid] = sc;
      }
    }
  };
  [[maybe_unused]] auto checkpoint = [&](const double* it) {
    double sr = 0.0, se = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      if (valid[s] && x >= own0 && x < own1) {
        const int64_t e = e0 + x;
        const int64_t row = e * M + i;
        const double* um = it + (x - 1) * M;
        const double* ux = it + x * M;
        const double* up = it + (x + 1) * M;
        double t = 0.0;
        if (CMP) {
          double dk[M], sc;
          row_entries(s, row, dk, sc);
          t += sc * um[a.lv.c_sub];
#pragma unroll
          for (int j = 0; j < M; ++j) t += dk[j] * ux[j];
          if (GRP) {
            const double d = group_sum<M>(qv[s][0] * up[i]);
            if (i == a.lv.r_sup) t += d;
          } else if (i == a.lv.r_sup) {
#pragma unroll
            for (int j = 0; j < (GRP ? 1 : M); ++j) t += qv[s][j] * up[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < M; ++j) t += a.lv.sub[row * M + j] * um[j];
#pragma unroll
          for (int j = 0; j < M; ++j) t += a.lv.dblk[row * M + j] * ux[j];
#pragma unroll
          for (int j = 0; j < M; ++j) t += a.lv.sup[row * M + j] * up[j];
        }
        const double r = bb[s] - t;
        sr += r * r;
      }
      __builtin_amdgcn_sched_barrier(0);   // one slab's row entries in flight at a time: the sweeps' registers stay live here
    }
    // (the error norm after the residual rows, not among them: fewer registers live at once)
    if (a.chk_exact) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int x = s * EPS + le;
        if (valid[s] && x >= own0 && x < own1) {
          const double d = uu[s] - a.chk_exact[(e0 + x) * M + i];
          se += d * d;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sr += __shfl_xor(sr, off, 64);
      se += __shfl_xor(se, off, 64);
    }
    stashed = true;
    double* red = lds + 2 * (TE + 2) * M;   // (the launch reserves 2 * NT / 64 doubles behind the iterate buffers)
    if ((tid & 63) == 0) {
      red[2 * (tid >> 6)] = sr;
      red[2 * (tid >> 6) + 1] = se;
    }
    __syncthreads();
    if (tid == 0) {
      double tr = 0.0, te = 0.0;
      for (int w = 0; w < NT / 64; ++w) {
        tr += red[2 * w];
        te += red[2 * w + 1];
      }
      double* out = a.chk_part + (kchk * a.chk_tiles + fused_tile(a)) * 2;
      out[0] = tr;
      out[1] = te;
    }
    ++kchk;
  };
  [[maybe_unused]] auto chk_due = [&](int sw) { return sw >= a.chk_sweep && (sw - a.chk_sweep) % a.chk_stride == 0; };
  for (int sw = 0; sw < nhalf; ++sw) {
    if constexpr (CHK) {
      if (chk_due(sw)) checkpoint(cur);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      if (active) {
        const double* um = cur + (x - 1) * M;
        const double* up = cur + (x + 1) * M;
        double acc = g[s];
        if (CMP) {
          acc -= pc[s] * um[a.lv.c_sub];
          double dot = 0.0;
          if (GRP) {
            dot = group_sum<M>(qv[s][0] * up[i]);
          } else {
#pragma unroll
            for (int j = 0; j < (GRP ? 1 : M); ++j) dot += qv[s][j] * up[j];
          }
          acc -= binv_r[s] * dot;
        } else {
#pragma unroll
          for (int j = 0; j < M; ++j) acc -= Pr[s][j] * um[j];
#pragma unroll
          for (int j = 0; j < M; ++j) acc -= Qr[s][j] * up[j];
        }
        double un = uu[s] + a.alpha * (acc - uu[s]);
        if (!valid[s]) un = 0.0;
        if constexpr (!GS) {
          uu[s] = un;
          nxt[x * M + i] = un;
        } else {
          const int colour = (sw & 1) ^ (a.gs == 2 ? 1 : 0);
          if (((e0 + x) & 1) == colour) {
            uu[s] = un;
            cur[x * M + i] = un;
          }
        }
      }
    }
    __syncthreads();
    if constexpr (!GS) {
      double* t = cur;
      cur = nxt;
      nxt = t;
    }
  }

  if constexpr (CHK) {
    if (a.chk_final || chk_due(nhalf)) checkpoint(cur);   // a launch that ends on a checked iterate: the last ascent
  }

  // ---- store the iterate of the owned elements ---------------------------------------------
  const int xo0 = own0, xo1 = own1;
  if (a.u_out) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      if (valid[s] && x >= xo0 && x < xo1) AGGMG_ST(a.u_out[(e0 + x) * M + i], uu[s]);
    }
  }

  if (!a.do_residual) return;

  double rr[NS];
  const double* lfo = a.lf_out;
  if (a.r_out || a.lf_out) {
    // ---- explicit residual r = b - A u on the owned elements, ascending column order ---------
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      rr[s] = 0.0;
      if (valid[s] && x >= xo0 && x < xo1) {
        const int64_t e = e0 + x;
        const int64_t row = e * M + i;
        const double* um = cur + (x - 1) * M;
        const double* ux = cur + x * M;
        const double* up = cur + (x + 1) * M;
        double t = 0.0;
        if (CMP) {
          double dk[M], sc;
          if constexpr (CHK) {
            row_entries(s, row, dk, sc);
          } else {
            sc = a.lv.scol[row];
#pragma unroll
            for (int j = 0; j < M; ++j) dk[j] = a.lv.dblk[row * M + j];
          }
          t += sc * um[a.lv.c_sub];
#pragma unroll
          for (int j = 0; j < M; ++j) t += dk[j] * ux[j];
          if (GRP) {
            const double d = group_sum<M>(qv[s][0] * up[i]);
            if (i == a.lv.r_sup) t += d;
          } else if (i == a.lv.r_sup) {
#pragma unroll
            for (int j = 0; j < (GRP ? 1 : M); ++j) t += qv[s][j] * up[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < M; ++j) t += a.lv.sub[row * M + j] * um[j];
#pragma unroll
          for (int j = 0; j < M; ++j) t += a.lv.dblk[row * M + j] * ux[j];
#pragma unroll
          for (int j = 0; j < M; ++j) t += a.lv.sup[row * M + j] * up[j];
        }
        rr[s] = bb[s] - t;
        if (a.r_out) a.r_out[row] = rr[s];
      }
    }
  } else if (a.ld_out) {
    // ---- preconditioned residual w = B^{-1}(b - A u) = g - P u- - u - Q u+ : what one more
    // sweep would add (before alpha); restricted with the precomputed (L'D) rows, so neither the
    // diagonal blocks nor L are read
    lfo = a.ld_out;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      rr[s] = 0.0;
      if (valid[s] && x >= xo0 && x < xo1) {
        const double* um = cur + (x - 1) * M;
        const double* up = cur + (x + 1) * M;
        double acc = g[s];
        if (CMP) {
          acc -= pc[s] * um[a.lv.c_sub];
          double dot = 0.0;
          if (GRP) {
            dot = group_sum<M>(qv[s][0] * up[i]);
          } else {
#pragma unroll
            for (int j = 0; j < (GRP ? 1 : M); ++j) dot += qv[s][j] * up[j];
          }
          acc -= binv_r[s] * dot;
        } else {
#pragma unroll
          for (int j = 0; j < M; ++j) acc -= Pr[s][j] * um[j];
#pragma unroll
          for (int j = 0; j < M; ++j) acc -= Qr[s][j] * up[j];
        }
        rr[s] = acc - uu[s];
      }
    }
  }
  if (!lfo) return;

  if (a.par_out) {
    // ---- restriction onto agglomerates of different sizes: r through LDS, one thread per (J, mode) over the
    // part of the agglomerate this tile owns
    const int mc = a.mc_out;
    const int64_t eo0 = e0 + xo0 > 0 ? e0 + xo0 : 0;
    const int64_t eo1 = e0 + xo1 < ne ? e0 + xo1 : ne;
    if (pre2) {
      // two coarse modes: the rows of L' came in with the tile's other streams; every row thread leaves its two
      // products in the (now free) iterate buffers and the (J, mode) threads only add
      __syncthreads();
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int x = s * EPS + le;
        const bool own = valid[s] && x >= xo0 && x < xo1;
        if (active) {
          nxt[x * M + i] = own ? l2x[s] * rr[s] : 0.0;
          cur[x * M + i] = own ? l2y[s] * rr[s] : 0.0;
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int x = s * EPS + le;
        if (active) nxt[x * M + i] = rr[s];
      }
    }
    __syncthreads();
    if (eo1 <= eo0) return;
    const int64_t Ja = a.par_out[eo0], Jb = a.par_out[eo1 - 1];
    for (int64_t t = tid; t < (Jb - Ja + 1) * mc; t += NT) {
      const int64_t J = Ja + t / mc;
      const int c = (int)(t - (J - Ja) * mc);
      const int64_t f0 = a.first_out[J], f1 = a.first_out[J + 1];
      const int64_t lo = f0 > eo0 ? f0 : eo0, hi = f1 < eo1 ? f1 : eo1;
      double acc = 0.0;
      if (pre2) {
        const double* pr = c ? cur : nxt;
        for (int64_t k = lo * M; k < hi * M; ++k) acc += pr[k - e0 * M];                    // ascending fine row
      } else {
        for (int64_t k = lo * M; k < hi * M; ++k) acc += lfo[k * mc + c] * nxt[k - e0 * M];  // ascending fine row
      }
      if (lo == f0 && hi == f1)  // (always, with the owned range on agglomerate boundaries)
        a.rc_out[J * mc + c] = acc;
      else
        atomicAdd(&a.rc_out[J * mc + c], acc);
    }
    return;
  }

  const int rho = a.rho_out, mc = a.mc_out;
  const int ncoarse = a.owned / rho;  // owned coarse elements of this tile
  const int64_t J0 = (fused_tile(a) * a.owned) / rho;
  const int64_t nec = ne / rho;
  if (pre2) {
    // ---- restriction, two coarse modes: every row thread forms its two products with the
    // preloaded L' entries; both iterate buffers are free once all threads are past the residual
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      const bool own = valid[s] && x >= xo0 && x < xo1;
      if (active) {
        nxt[x * M + i] = own ? l2x[s] * rr[s] : 0.0;
        cur[x * M + i] = own ? l2y[s] * rr[s] : 0.0;
      }
    }
    __syncthreads();
    for (int t = tid; t < ncoarse * 2; t += NT) {
      const int Jl = t >> 1, c = t & 1;
      const int64_t J = J0 + Jl;
      if (J >= nec) continue;
      const double* pr = (c ? cur : nxt) + (a.halo_left + Jl * rho) * M;
      double acc = 0.0;
      for (int k = 0; k < rho * M; ++k) acc += pr[k];  // ascending fine row, as the column dot of L'
      a.rc_out[J * 2 + c] = acc;
    }
    return;
  }
  // ---- restriction rc = L' r, general mode count: r through LDS, one thread per (J, mode) ----
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int x = s * EPS + le;
    if (active) nxt[x * M + i] = rr[s];
  }
  __syncthreads();
  for (int t = tid; t < ncoarse * mc; t += NT) {
    const int Jl = t / mc, c = t - Jl * mc;
    const int64_t J = J0 + Jl;
    if (J >= nec) continue;
    const int xb = a.halo_left + Jl * rho;       // first fine element (tile-local)
    const int64_t rowb = (e0 + xb) * (int64_t)M;  // first fine row (global)
    double acc = 0.0;
    for (int k = 0; k < rho * M; ++k) acc += lfo[(rowb + k) * mc + c] * nxt[xb * M + k];
    a.rc_out[J * mc + c] = acc;
  }
}

}  // namespace aggmg
