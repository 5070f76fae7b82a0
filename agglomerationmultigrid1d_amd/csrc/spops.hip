// Sparse set-up operations of the hierarchy constructors on the device:
//   BlockDiagonal * sparse, BlockDiagonalLU \ sparse     src/block_diagonal.jl:195-264, 314-383
//   sparse * sparse, sparse - sparse                     the `L'*X*L` and `C - D*(M_LU \ G)` of
//                                                        src/mesh_heirarchy.jl:71-72,79-84,98-103
// All operate column by column on CSC arrays (one thread per result column; 1-D finite-element
// matrices have short columns) and produce CSC, so results are ordinary operators: they can be
// smoothed, restricted with, downloaded.  Accumulation order follows SparseArrays: for a product the
// entries of B's column in ascending row order, inside each the entries of A's column in ascending
// row order.  Compiled with -ffp-contract=off.
// DEVIATION from the reference's operation sequence: `BlockDiagonalLU \ sparse` multiplies by the explicit inverse
// of every block's pivoted LU (K6, setup_invert_blocks) where bd_sp_colsolve runs `mBlocksLU[blockInd] \ tempVec`
// (getrs, src/block_diagonal.jl:374) -- equal to round-off for the well-conditioned mass blocks this is used on
// (condition numbers O(10)), pinned against the oracle at 1e-12, not against an executed reference.
#include <hipcub/hipcub.hpp>

#include "internal.hpp"

namespace {

constexpr int kSetupThreads = 256;
constexpr int kColCap = 128;  // longest result column the product kernels hold in thread-local storage

inline unsigned grid_for(int64_t n) { return (unsigned)((n + kSetupThreads - 1) / kSetupThreads); }

// ---- BlockDiagonal (or its inverse) times a sparse matrix --------------------------------------------
// count: distinct blocks touched by every column of S, times m
__global__ __launch_bounds__(kSetupThreads) void bdsp_count_kernel(int64_t ncols, int m, const int32_t* __restrict__ colptr,
                                                                   const int32_t* __restrict__ rowval,
                                                                   int32_t* __restrict__ counts) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j >= ncols) return;
  int64_t last = -1;
  int32_t cnt = 0;
  for (int32_t p = colptr[j]; p < colptr[j + 1]; ++p) {
    const int64_t b = rowval[p] / m;
    if (b != last) {
      cnt += m;
      last = b;
    }
  }
  counts[j] = cnt;
}

// fill: for every touched block the gathered dense sub-vector times the block matrix; ALL m rows emitted
__global__ __launch_bounds__(kSetupThreads) void bdsp_fill_kernel(int64_t ncols, int m, const double* __restrict__ mats,
                                                                  const int32_t* __restrict__ colptr,
                                                                  const int32_t* __restrict__ rowval,
                                                                  const double* __restrict__ vals,
                                                                  const int32_t* __restrict__ ocolptr,
                                                                  int32_t* __restrict__ orowval, double* __restrict__ ovals) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j >= ncols) return;
  int32_t o = ocolptr[j];
  int32_t p = colptr[j];
  const int32_t p1 = colptr[j + 1];
  double t[64];
  while (p < p1) {
    const int64_t b = rowval[p] / m;
    for (int i = 0; i < m; ++i) t[i] = 0.0;
    while (p < p1 && rowval[p] / m == b) {
      t[rowval[p] - b * m] = vals[p];
      ++p;
    }
    const double* Mb = mats + b * m * m;
    for (int i = 0; i < m; ++i) {
      double acc = 0.0;
      for (int k = 0; k < m; ++k) acc += Mb[i * m + k] * t[k];
      orowval[o] = (int32_t)(b * m + i);
      ovals[o] = acc;
      ++o;
    }
  }
}

// ---- C = A * B, all CSC ----------------------------------------------------------------------------
// Merge of the columns of A selected by column j of B into a sorted thread-local list.
// FILL = false: count the distinct rows; FILL = true: accumulate and write.  err[0] raised when a result
// column exceeds kColCap rows.
template <bool FILL>
__global__ __launch_bounds__(kSetupThreads) void spmm_kernel(int64_t ncols, const int32_t* __restrict__ acp,
                                                             const int32_t* __restrict__ arv,
                                                             const double* __restrict__ av,
                                                             const int32_t* __restrict__ bcp,
                                                             const int32_t* __restrict__ brv,
                                                             const double* __restrict__ bv, int32_t* __restrict__ counts,
                                                             const int32_t* __restrict__ ocolptr,
                                                             int32_t* __restrict__ orowval, double* __restrict__ ovals,
                                                             int* __restrict__ err) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j >= ncols) return;
  int32_t rows[kColCap];
  double acc[kColCap];
  int len = 0;
  for (int32_t q = bcp[j]; q < bcp[j + 1]; ++q) {
    const int32_t k = brv[q];
    const double bkj = bv[q];
    int pos = 0;  // both lists ascend: the search position only moves forward inside one column of A
    for (int32_t p = acp[k]; p < acp[k + 1]; ++p) {
      const int32_t r = arv[p];
      while (pos < len && rows[pos] < r) ++pos;
      if (pos == len || rows[pos] != r) {
        if (len >= kColCap) {
          err[0] = 1;
          return;
        }
        for (int s = len; s > pos; --s) {
          rows[s] = rows[s - 1];
          if (FILL) acc[s] = acc[s - 1];
        }
        rows[pos] = r;
        if (FILL) acc[pos] = 0.0;
        ++len;
      }
      if (FILL) acc[pos] += av[p] * bkj;
    }
  }
  if (!FILL) {
    counts[j] = len;
  } else {
    const int32_t o = ocolptr[j];
    for (int s = 0; s < len; ++s) {
      orowval[o + s] = rows[s];
      ovals[o + s] = acc[s];
    }
  }
}

// ---- C = A - B, numerically-zero results dropped -----------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(kSetupThreads) void spsub_kernel(int64_t ncols, const int32_t* __restrict__ acp,
                                                              const int32_t* __restrict__ arv,
                                                              const double* __restrict__ av,
                                                              const int32_t* __restrict__ bcp,
                                                              const int32_t* __restrict__ brv,
                                                              const double* __restrict__ bv, int32_t* __restrict__ counts,
                                                              const int32_t* __restrict__ ocolptr,
                                                              int32_t* __restrict__ orowval, double* __restrict__ ovals) {
  const int64_t j = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (j >= ncols) return;
  int32_t pa = acp[j], pb = bcp[j];
  const int32_t ea = acp[j + 1], eb = bcp[j + 1];
  int32_t o = FILL ? ocolptr[j] : 0;
  int32_t cnt = 0;
  while (pa < ea || pb < eb) {
    int32_t r;
    double v;
    if (pb >= eb || (pa < ea && arv[pa] < brv[pb])) {
      r = arv[pa];
      v = av[pa++];
    } else if (pa >= ea || brv[pb] < arv[pa]) {
      r = brv[pb];
      v = -bv[pb++];
    } else {
      r = arv[pa];
      v = av[pa++] - bv[pb++];
    }
    if (v != 0.0) {
      if (FILL) {
        orowval[o] = r;
        ovals[o] = v;
        ++o;
      }
      ++cnt;
    }
  }
  if (!FILL) counts[j] = cnt;
}

struct Tmp {
  void* p = nullptr;
  ~Tmp() {
    if (p) (void)hipFree(p);
  }
};

// int32 counts widened on the fly: the total of a product can pass 2^31 long before any single count does
struct WidenCount {
  __host__ __device__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

// exclusive scan of counts[0..n) into colptr[0..n] (colptr[n] = total).  The total is formed in 64 bits FIRST:
// an int32 scan of a result with >= 2^31 entries wraps, and the fill kernels would then write through wrapped
// offsets.  Returns AGGMG_ERR_UNSUPPORTED (nothing scanned) when the total does not fit int32 indices.
int scan_counts(aggmg_ctx* ctx, int32_t* counts_np1, int32_t* colptr, int64_t n, int64_t* total, const char* who) {
  hipcub::TransformInputIterator<int64_t, WidenCount, const int32_t*> wide(counts_np1, WidenCount());
  Tmp sum;
  HIPCHK(hipMalloc(&sum.p, sizeof(int64_t)));
  size_t rbytes = 0;
  HIPCHK(hipcub::DeviceReduce::Sum(nullptr, rbytes, wide, (int64_t*)sum.p, (int)n, ctx->stream));
  Tmp rt;
  HIPCHK(hipMalloc(&rt.p, std::max<size_t>(rbytes, 8)));
  HIPCHK(hipcub::DeviceReduce::Sum(rt.p, rbytes, wide, (int64_t*)sum.p, (int)n, ctx->stream));
  int64_t tot = 0;
  HIPCHK(hipMemcpyAsync(&tot, sum.p, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *total = tot;
  if (tot >= ((int64_t)1 << 31))
    return fail(ctx, AGGMG_ERR_UNSUPPORTED, std::string(who) + ": result has " + std::to_string(tot) + " >= 2^31 entries (int32 device indices)");
  size_t bytes = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, counts_np1, colptr, (int)(n + 1), ctx->stream));
  Tmp t;
  HIPCHK(hipMalloc(&t.p, std::max<size_t>(bytes, 8)));
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(t.p, bytes, counts_np1, colptr, (int)(n + 1), ctx->stream));
  return AGGMG_OK;
}

int new_op(aggmg_ctx* ctx, int64_t m, int64_t n, int kind, std::unique_ptr<aggmg_op>* out, int32_t** counts) {
  auto op = std::make_unique<aggmg_op>();
  op->m = m;
  op->n = n;
  op->kind = kind;
  op->csc.nrows = n;
  op->csc.ncols = m;
  HIPCHK(hipMalloc((void**)&op->csc.rowptr, (size_t)(n + 1) * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)counts, (size_t)(n + 1) * sizeof(int32_t)));
  HIPCHK(hipMemsetAsync(*counts, 0, (size_t)(n + 1) * sizeof(int32_t), ctx->stream));
  *out = std::move(op);
  return AGGMG_OK;
}

int alloc_entries(aggmg_ctx* ctx, aggmg_op* op, int64_t nnz) {
  op->nnz = nnz;
  op->csc.nnz = nnz;
  HIPCHK(hipMalloc((void**)&op->csc.colind, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&op->csc.vals, (size_t)std::max<int64_t>(nnz, 1) * sizeof(double)));
  return AGGMG_OK;
}

int check_kind(aggmg_ctx* ctx, int kind, const char* who) {
  if (kind != AGGMG_OP_STIFFNESS && kind != AGGMG_OP_TRANSFER) return fail(ctx, AGGMG_ERR_ARGUMENT, std::string(who) + ": unknown kind");
  return AGGMG_OK;
}

}  // namespace

extern "C" int aggmg_bd_sp_apply(aggmg_ctx* ctx, aggmg_smoother* bd, aggmg_op* S, int kind, aggmg_op** out) {
  if (!ctx || !bd || !S || !out) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_bd_sp_apply: NULL argument");
  *out = nullptr;
  CHECK(check_kind(ctx, kind, "aggmg_bd_sp_apply"));
  if (!bd->contiguous || bd->overlapping || !bd->binv || bd->m > 64)
    return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_bd_sp_apply: needs a block object with contiguous aligned blocks (aggmg_blockdiag_setup)");
  if (bd->N != S->m) return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_bd_sp_apply: DimensionMismatch (block_diagonal.jl:196,315)");
  HIPCHK(hipSetDevice(ctx->device));
  const int m = (int)bd->m;
  std::unique_ptr<aggmg_op> op;
  int32_t* counts = nullptr;
  CHECK(new_op(ctx, S->m, S->n, kind, &op, &counts));
  Tmp cown;
  cown.p = counts;
  const int64_t nc = S->n;
  if (nc) hipLaunchKernelGGL(bdsp_count_kernel, dim3(grid_for(nc)), dim3(kSetupThreads), 0, ctx->stream, nc, m,
                             (const int32_t*)S->csc.rowptr, (const int32_t*)S->csc.colind, counts);
  HIPCHK(hipGetLastError());
  int64_t nnz = 0;
  CHECK(scan_counts(ctx, counts, op->csc.rowptr, nc, &nnz, "aggmg_bd_sp_apply"));
  CHECK(alloc_entries(ctx, op.get(), nnz));
  if (nc) hipLaunchKernelGGL(bdsp_fill_kernel, dim3(grid_for(nc)), dim3(kSetupThreads), 0, ctx->stream, nc, m,
                             (const double*)bd->binv, (const int32_t*)S->csc.rowptr, (const int32_t*)S->csc.colind,
                             (const double*)S->csc.vals, (const int32_t*)op->csc.rowptr, op->csc.colind, op->csc.vals);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *out = op.release();
  return AGGMG_OK;
}

extern "C" int aggmg_sp_matmul(aggmg_ctx* ctx, aggmg_op* A, aggmg_op* B, int kind, aggmg_op** out) {
  if (!ctx || !A || !B || !out) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_sp_matmul: NULL argument");
  *out = nullptr;
  CHECK(check_kind(ctx, kind, "aggmg_sp_matmul"));
  if (A->n != B->m) return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_sp_matmul: DimensionMismatch");
  HIPCHK(hipSetDevice(ctx->device));
  std::unique_ptr<aggmg_op> op;
  int32_t* counts = nullptr;
  CHECK(new_op(ctx, A->m, B->n, kind, &op, &counts));
  Tmp cown;
  cown.p = counts;
  int* err = nullptr;
  HIPCHK(hipMalloc((void**)&err, sizeof(int)));
  Tmp eown;
  eown.p = err;
  HIPCHK(hipMemsetAsync(err, 0, sizeof(int), ctx->stream));
  const int64_t nc = B->n;
  const int32_t *acp = A->csc.rowptr, *arv = A->csc.colind, *bcp = B->csc.rowptr, *brv = B->csc.colind;
  const double *av = A->csc.vals, *bv = B->csc.vals;
  if (nc) hipLaunchKernelGGL((spmm_kernel<false>), dim3(grid_for(nc)), dim3(kSetupThreads), 0, ctx->stream, nc, acp, arv, av, bcp, brv,
                             bv, counts, (const int32_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, err);
  HIPCHK(hipGetLastError());
  int herr = 0;
  HIPCHK(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  int64_t nnz = 0;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (herr)
    return fail(ctx, AGGMG_ERR_UNSUPPORTED, "aggmg_sp_matmul: a result column has more than " + std::to_string(kColCap) +
                                                " rows (the device product is meant for finite-element matrices with short columns)");
  CHECK(scan_counts(ctx, counts, op->csc.rowptr, nc, &nnz, "aggmg_sp_matmul"));
  CHECK(alloc_entries(ctx, op.get(), nnz));
  if (nc) hipLaunchKernelGGL((spmm_kernel<true>), dim3(grid_for(nc)), dim3(kSetupThreads), 0, ctx->stream, nc, acp, arv, av, bcp, brv,
                             bv, counts, (const int32_t*)op->csc.rowptr, op->csc.colind, op->csc.vals, err);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *out = op.release();
  return AGGMG_OK;
}

extern "C" int aggmg_sp_sub(aggmg_ctx* ctx, aggmg_op* A, aggmg_op* B, int kind, aggmg_op** out) {
  if (!ctx || !A || !B || !out) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_sp_sub: NULL argument");
  *out = nullptr;
  CHECK(check_kind(ctx, kind, "aggmg_sp_sub"));
  if (A->m != B->m || A->n != B->n) return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_sp_sub: DimensionMismatch");
  HIPCHK(hipSetDevice(ctx->device));
  std::unique_ptr<aggmg_op> op;
  int32_t* counts = nullptr;
  CHECK(new_op(ctx, A->m, A->n, kind, &op, &counts));
  Tmp cown;
  cown.p = counts;
  const int64_t nc = A->n;
  const int32_t *acp = A->csc.rowptr, *arv = A->csc.colind, *bcp = B->csc.rowptr, *brv = B->csc.colind;
  const double *av = A->csc.vals, *bv = B->csc.vals;
  if (nc) hipLaunchKernelGGL((spsub_kernel<false>), dim3(grid_for(nc)), dim3(kSetupThreads), 0, ctx->stream, nc, acp, arv, av, bcp, brv,
                             bv, counts, (const int32_t*)nullptr, (int32_t*)nullptr, (double*)nullptr);
  HIPCHK(hipGetLastError());
  int64_t nnz = 0;
  CHECK(scan_counts(ctx, counts, op->csc.rowptr, nc, &nnz, "aggmg_sp_sub"));
  CHECK(alloc_entries(ctx, op.get(), nnz));
  if (nc) hipLaunchKernelGGL((spsub_kernel<true>), dim3(grid_for(nc)), dim3(kSetupThreads), 0, ctx->stream, nc, acp, arv, av, bcp, brv,
                             bv, counts, (const int32_t*)op->csc.rowptr, op->csc.colind, op->csc.vals);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *out = op.release();
  return AGGMG_OK;
}

// Test aid: the count -> column-pointer scan of the set-up products on caller-supplied counts (host array), so that
// the 2^31 guard can be exercised without building a 2^31-entry product.
extern "C" int aggmg_debug_scan_counts(aggmg_ctx* ctx, const int32_t* counts_host, int64_t n, int64_t* total) {
  if (!ctx || !counts_host || !total || n < 0 || n >= ((int64_t)1 << 31) - 1) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_debug_scan_counts: bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  Tmp c, p;
  HIPCHK(hipMalloc(&c.p, (size_t)(n + 1) * sizeof(int32_t)));
  HIPCHK(hipMalloc(&p.p, (size_t)(n + 1) * sizeof(int32_t)));
  HIPCHK(hipMemsetAsync(c.p, 0, (size_t)(n + 1) * sizeof(int32_t), ctx->stream));
  if (n) HIPCHK(hipMemcpyAsync(c.p, counts_host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  return scan_counts(ctx, (int32_t*)c.p, (int32_t*)p.p, n, total, "aggmg_debug_scan_counts");
}

// the transposed operator as an operator of its own (L' of L'*X*L): its CSC arrays are the row-gather CSR of
// the original, built by the device transposition
extern "C" int aggmg_op_transpose(aggmg_ctx* ctx, aggmg_op* A, int kind, aggmg_op** out) {
  if (!ctx || !A || !out) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_op_transpose: NULL argument");
  *out = nullptr;
  CHECK(check_kind(ctx, kind, "aggmg_op_transpose"));
  HIPCHK(hipSetDevice(ctx->device));
  CHECK(op_ensure_csr(ctx, A));
  auto op = std::make_unique<aggmg_op>();
  op->m = A->n;
  op->n = A->m;
  op->nnz = A->nnz;
  op->kind = kind;
  op->csc.nrows = A->m;
  op->csc.ncols = A->n;
  op->csc.nnz = A->nnz;
  HIPCHK(hipMalloc((void**)&op->csc.rowptr, (size_t)(A->m + 1) * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&op->csc.colind, (size_t)std::max<int64_t>(A->nnz, 1) * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&op->csc.vals, (size_t)std::max<int64_t>(A->nnz, 1) * sizeof(double)));
  HIPCHK(hipMemcpyAsync(op->csc.rowptr, A->csr.rowptr, (size_t)(A->m + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
  if (A->nnz) {
    HIPCHK(hipMemcpyAsync(op->csc.colind, A->csr.colind, (size_t)A->nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(op->csc.vals, A->csr.vals, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *out = op.release();
  return AGGMG_OK;
}

// CSC arrays of an operator back to the host (0-based int32): parity tests of the set-up products
extern "C" int aggmg_op_download_csc(aggmg_ctx* ctx, const aggmg_op* op, int32_t* colptr, int32_t* rowval, double* nzval) {
  if (!ctx || !op) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_op_download_csc: NULL");
  const CsrDev& d = op->csc;
  if (colptr) HIPCHK(hipMemcpyAsync(colptr, d.rowptr, (size_t)(op->n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (rowval && op->nnz) HIPCHK(hipMemcpyAsync(rowval, d.colind, (size_t)op->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (nzval && op->nnz) HIPCHK(hipMemcpyAsync(nzval, d.vals, (size_t)op->nnz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return AGGMG_OK;
}

// inverses (or matrices) of a block object, [nb][m][m] row-major: evidence for the device LU (K6)
extern "C" int aggmg_smoother_download_blocks(aggmg_ctx* ctx, const aggmg_smoother* sm, double* out) {
  if (!ctx || !sm || !out) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_smoother_download_blocks: NULL");
  if (!sm->binv) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_smoother_download_blocks: not a block smoother");
  const int64_t cnt = sm->nb * sm->m * sm->m;
  if (cnt) HIPCHK(hipMemcpyAsync(out, sm->binv, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return AGGMG_OK;
}
