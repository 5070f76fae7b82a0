// Device-side set-up of libaggmg_hip: host code here only allocates, launches the
// kernels of setup_kernels.hpp and reads back a handful of flags -- no O(n) host loop touches an
// operator any more.  Compiled with -ffp-contract=off (see setup_kernels.hpp).
#include <hipcub/hipcub.hpp>

#include "internal.hpp"
#include "setup_kernels.hpp"

namespace {

inline unsigned grid_for(int64_t n) { return (unsigned)((n + kSetupThreads - 1) / kSetupThreads); }

// device scratch that frees itself
struct Tmp {
  void* p = nullptr;
  ~Tmp() {
    if (p) (void)hipFree(p);
  }
  template <typename T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

int tmp_alloc(aggmg_ctx* ctx, Tmp* t, size_t bytes, bool zero) {
  HIPCHK(hipMalloc(&t->p, std::max<size_t>(bytes, 8)));
  if (zero) HIPCHK(hipMemsetAsync(t->p, 0, std::max<size_t>(bytes, 8), ctx->stream));
  return AGGMG_OK;
}

template <typename T>
int dalloc(aggmg_ctx* ctx, T** out, int64_t count, bool zero) {
  *out = nullptr;
  const size_t bytes = (size_t)std::max<int64_t>(count, 1) * sizeof(T);
  HIPCHK(hipMalloc((void**)out, bytes));
  if (zero) HIPCHK(hipMemsetAsync(*out, 0, bytes, ctx->stream));
  return AGGMG_OK;
}

// a few ints of flags on the device, read back synchronously
struct Flags {
  int* d = nullptr;
  int n = 0;
  ~Flags() {
    if (d) (void)hipFree(d);
  }
  int init(aggmg_ctx* ctx, int count) {
    n = count;
    HIPCHK(hipMalloc((void**)&d, count * sizeof(int)));
    HIPCHK(hipMemsetAsync(d, 0, count * sizeof(int), ctx->stream));
    return AGGMG_OK;
  }
  int read(aggmg_ctx* ctx, int* host) {
    HIPCHK(hipMemcpyAsync(host, d, n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return AGGMG_OK;
  }
  int clear(aggmg_ctx* ctx) {
    HIPCHK(hipMemsetAsync(d, 0, n * sizeof(int), ctx->stream));
    return AGGMG_OK;
  }
};

#define LAUNCH(kern, n, ...)                                                                      \
  do {                                                                                            \
    if ((n) > 0) hipLaunchKernelGGL(kern, dim3(grid_for(n)), dim3(kSetupThreads), 0, ctx->stream, __VA_ARGS__); \
    HIPCHK(hipGetLastError());                                                                    \
  } while (0)

}  // namespace

// ---------------------------------------------------------------------------------------------
// upload
// ---------------------------------------------------------------------------------------------
int setup_csc_upload(aggmg_ctx* ctx, int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval,
                     const double* nzval, int one_based, CsrDev* out) {
  const int64_t base = one_based ? 1 : 0;
  const int64_t nnz = colptr[n] - base;
  out->nrows = n;  // rows of the transposed orientation = columns of the matrix
  out->ncols = m;
  out->nnz = nnz;
  CHECK(dalloc(ctx, &out->rowptr, n + 1, false));
  CHECK(dalloc(ctx, &out->colind, nnz, false));
  CHECK(dalloc(ctx, &out->vals, nnz, false));
  Tmp cp64, rv64;
  CHECK(tmp_alloc(ctx, &cp64, (size_t)(n + 1) * 8, false));
  CHECK(tmp_alloc(ctx, &rv64, (size_t)std::max<int64_t>(nnz, 1) * 8, false));
  HIPCHK(hipMemcpyAsync(cp64.p, colptr, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  if (nnz) {
    HIPCHK(hipMemcpyAsync(rv64.p, rowval, (size_t)nnz * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(out->vals, nzval, (size_t)nnz * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  Flags f;
  CHECK(f.init(ctx, 4));
  LAUNCH(csc_convert_colptr_kernel, n + 1, n, cp64.as<int64_t>(), base, nnz, out->rowptr, f.d);
  int h[4];
  CHECK(f.read(ctx, h));  // the row pass below walks colptr: it must be sane first
  if (h[0]) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_csc_upload: colptr not monotone");
  LAUNCH(csc_convert_rows_kernel, n, n, m, (const int32_t*)out->rowptr, rv64.as<int64_t>(), base, out->colind, f.d);
  CHECK(f.read(ctx, h));
  if (h[1]) return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_csc_upload: row index out of range");
  if (h[2]) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_csc_upload: row indices not strictly ascending in a column");
  return AGGMG_OK;
}

// CSR-stream row blocks of a device CSR (rows cut into runs of <= kStreamNnz entries and <= 4 * kThreads
// rows); only the generic kernels need them
int setup_stream_blocks(aggmg_ctx* ctx, CsrDev* d) {
  if (d->rowblk || d->nrows == 0) return AGGMG_OK;
  d->lpr = 1;
  {
    const double avg = (double)d->nnz / (double)d->nrows;
    while (d->lpr < 64 && (double)d->lpr * 1.5 < avg) d->lpr *= 2;
  }
  if ((double)d->nnz / (double)d->nrows > 48.0) return AGGMG_OK;  // long rows: lanes-per-row kernel
  std::vector<int32_t> rowptr(d->nrows + 1);
  HIPCHK(hipMemcpyAsync(rowptr.data(), d->rowptr, rowptr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  {
    int32_t mx = 0;
    for (int64_t i = 0; i < d->nrows; ++i) mx = std::max(mx, rowptr[i + 1] - rowptr[i]);
    d->maxrow = mx;
  }
  const int64_t nrows = d->nrows;
  const std::vector<int32_t> blk = stream_row_blocks(rowptr.data(), nrows, kStreamNnz, kStreamRows);   // host_plan.hpp
  d->nblk = (int64_t)blk.size() - 1;
  CHECK(dev_upload(ctx, blk, &d->rowblk));
  // square and banded: row blocks for csr_band_kernel (x window in LDS, several point-Jacobi sweeps per launch) --
  // a tile -- a block's rows plus (S - 1) * bw halo rows on either side -- has at most kThreads rows and kBandNnz entries;
  // S, the most sweeps per launch, is chosen per operator so that the halo stays near a quarter of the tile
  static const bool band_on = [] {
    const char* e = std::getenv("AGGMG_CSR_BAND");
    return !(e && e[0] == '0');
  }();
  if (band_on && d->nrows == d->ncols && nrows > 1) {
    Flags f;
    CHECK(f.init(ctx, 1));
    LAUNCH(csr_bandwidth_kernel, nrows, d->view(), f.d);
    int bwv = 0;
    CHECK(f.read(ctx, &bwv));
    if (bwv <= kBandMaxBw) {
      const int bw = std::max(bwv, 1);
      std::vector<int32_t> bb;
      const int S = std::max(2, std::min(kBandSweeps, 1 + 32 / bw));
      // (band_row_blocks: block rows + 2 * S * bw <= window  <=>  tile rows = block rows + 2 (S - 1) bw <= kThreads)
      const bool ok = band_row_blocks(rowptr.data(), nrows, bw, S, kBandNnz, kThreads + 2 * bw, kThreads, &bb);   // host_plan.hpp
      if (ok) {
        d->band_sweeps = S;
        d->bw = bw;
        d->nbandblk = (int64_t)bb.size() - 1;
        CHECK(dev_upload(ctx, bb, &d->bandblk));
      }
    }
  }
  return AGGMG_OK;
}

// row-gather CSR of the matrix from its CSC arrays: stable radix sort of the entries by row
int setup_transpose(aggmg_ctx* ctx, const CsrDev& csc, int64_t m, CsrDev* out) {
  const int64_t n = csc.nrows, nnz = csc.nnz;
  out->nrows = m;
  out->ncols = n;
  out->nnz = nnz;
  CHECK(dalloc(ctx, &out->rowptr, m + 1, true));
  CHECK(dalloc(ctx, &out->colind, nnz, false));
  CHECK(dalloc(ctx, &out->vals, nnz, false));
  if (nnz == 0) return AGGMG_OK;
  // rowptr: histogram of the row indices, exclusive scan
  Tmp counts;
  CHECK(tmp_alloc(ctx, &counts, (size_t)(m + 1) * 4, true));
  LAUNCH(row_count_kernel, nnz, nnz, (const int32_t*)csc.colind, counts.as<int32_t>());
  size_t sbytes = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, sbytes, counts.as<int32_t>(), out->rowptr, (int)(m + 1), ctx->stream));
  Tmp stmp;
  CHECK(tmp_alloc(ctx, &stmp, sbytes, false));
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(stmp.p, sbytes, counts.as<int32_t>(), out->rowptr, (int)(m + 1), ctx->stream));
  // entries sorted by row; the sort is stable, so columns stay ascending inside a row
  Tmp ecol, iota, perm, keys;
  CHECK(tmp_alloc(ctx, &ecol, (size_t)nnz * 4, false));
  CHECK(tmp_alloc(ctx, &iota, (size_t)nnz * 4, false));
  CHECK(tmp_alloc(ctx, &perm, (size_t)nnz * 4, false));
  CHECK(tmp_alloc(ctx, &keys, (size_t)nnz * 4, false));
  LAUNCH(csc_entry_cols_kernel, n, n, (const int32_t*)csc.rowptr, ecol.as<int32_t>());
  LAUNCH(iota_kernel, nnz, nnz, iota.as<uint32_t>());
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < m) ++bits;
  size_t rbytes = 0;
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, rbytes, (const uint32_t*)csc.colind, keys.as<uint32_t>(), iota.as<uint32_t>(),
                                            perm.as<uint32_t>(), (int)nnz, 0, bits, ctx->stream));
  Tmp rtmp;
  CHECK(tmp_alloc(ctx, &rtmp, rbytes, false));
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(rtmp.p, rbytes, (const uint32_t*)csc.colind, keys.as<uint32_t>(), iota.as<uint32_t>(),
                                            perm.as<uint32_t>(), (int)nnz, 0, bits, ctx->stream));
  LAUNCH(csr_gather_kernel, nnz, nnz, (const uint32_t*)perm.as<uint32_t>(), (const int32_t*)ecol.as<int32_t>(),
         (const double*)csc.vals, out->colind, out->vals);
  HIPCHK(hipStreamSynchronize(ctx->stream));  // the temporaries go out of scope
  return AGGMG_OK;
}

int op_ensure_csr(aggmg_ctx* ctx, aggmg_op* op) {
  if (!op->csr.rowptr) CHECK(setup_transpose(ctx, op->csc, op->m, &op->csr));
  return setup_stream_blocks(ctx, &op->csr);
}

int op_ensure_csc_blocks(aggmg_ctx* ctx, aggmg_op* op) { return setup_stream_blocks(ctx, &op->csc); }

// host copy of the row-gather CSR (the host banded LU fallback of the coarsest solve wants one)
int op_host_csr(aggmg_ctx* ctx, aggmg_op* op, HostCsr* h) {
  CHECK(op_ensure_csr(ctx, op));
  h->rowptr.resize(op->m + 1);
  h->colind.resize(op->nnz);
  h->vals.resize(op->nnz);
  HIPCHK(hipMemcpyAsync(h->rowptr.data(), op->csr.rowptr, (op->m + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (op->nnz) {
    HIPCHK(hipMemcpyAsync(h->colind.data(), op->csr.colind, op->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(h->vals.data(), op->csr.vals, op->nnz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return AGGMG_OK;
}

// ---------------------------------------------------------------------------------------------
// smoothers
// ---------------------------------------------------------------------------------------------
int setup_jacobi_diag(aggmg_ctx* ctx, const aggmg_op* A, double** diag) {
  CHECK(dalloc(ctx, diag, A->m, false));
  LAUNCH(diag_extract_kernel, A->m, A->m, (const int32_t*)A->csc.rowptr, (const int32_t*)A->csc.colind,
         (const double*)A->csc.vals, *diag);
  return AGGMG_OK;
}

// row -> covering entries of the index lists (flat index block * m + i), rows by histogram + scan, the entries by a
// stable sort on the row: ascending flat index inside a row, the same order every time
int setup_block_order(aggmg_ctx* ctx, aggmg_smoother* sm) {
  if (sm->ordered) return AGGMG_OK;
  const int64_t N = sm->N, total = sm->nb * sm->m;
  // The one-pass sweep reads the operator rows of a block where they lie: blocks listed in an order unrelated to
  // their indices would turn that into scattered 100-byte reads (measured: 498 us per sweep against 289 us for the
  // four-launch form on 2^20 element blocks listed in random order).  The order of the blocks means nothing to an
  // additive / hybrid smoother, so they are put in ascending order of their smallest index, once, in place.
  if (sm->nb > 1) {
    Flags uns;
    CHECK(uns.init(ctx, 1));
    Tmp keys, keys2, iota, order;
    CHECK(tmp_alloc(ctx, &keys, (size_t)sm->nb * 4, false));
    LAUNCH(block_minkey_kernel, sm->nb, sm->nb, (int)sm->m, (const int32_t*)sm->inds, keys.as<uint32_t>(), uns.d);
    int u1 = 0;
    CHECK(uns.read(ctx, &u1));
    if (u1) {
      CHECK(tmp_alloc(ctx, &keys2, (size_t)sm->nb * 4, false));
      CHECK(tmp_alloc(ctx, &iota, (size_t)sm->nb * 4, false));
      CHECK(tmp_alloc(ctx, &order, (size_t)sm->nb * 4, false));
      LAUNCH(iota_kernel, sm->nb, sm->nb, iota.as<uint32_t>());
      int kb = 1;
      while (kb < 32 && ((int64_t)1 << kb) < N) ++kb;
      size_t rb = 0;
      HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, rb, keys.as<uint32_t>(), keys2.as<uint32_t>(), iota.as<uint32_t>(),
                                                order.as<uint32_t>(), (int)sm->nb, 0, kb, ctx->stream));
      Tmp rt;
      CHECK(tmp_alloc(ctx, &rt, rb, false));
      HIPCHK(hipcub::DeviceRadixSort::SortPairs(rt.p, rb, keys.as<uint32_t>(), keys2.as<uint32_t>(), iota.as<uint32_t>(),
                                                order.as<uint32_t>(), (int)sm->nb, 0, kb, ctx->stream));
      int32_t* inds2 = nullptr;
      double* binv2 = nullptr;
      CHECK(dalloc(ctx, &inds2, total, false));
      Tmp o1;
      o1.p = inds2;
      CHECK(dalloc(ctx, &binv2, total * sm->m, false));
      Tmp o2;
      o2.p = binv2;
      LAUNCH(block_permute_kernel, total, total, (int)sm->m, (const uint32_t*)order.as<uint32_t>(), (const int32_t*)sm->inds,
             (const double*)sm->binv, inds2, binv2);
      HIPCHK(hipStreamSynchronize(ctx->stream));
      o1.p = sm->inds;   // the old arrays are released with the temporaries
      o2.p = sm->binv;
      sm->inds = inds2;
      sm->binv = binv2;
    }
  }
  sm->ordered = true;
  return AGGMG_OK;
}

int setup_block_cover(aggmg_ctx* ctx, aggmg_smoother* sm) {
  if (sm->cover_ptr) return AGGMG_OK;
  CHECK(setup_block_order(ctx, sm));
  const int64_t N = sm->N, total = sm->nb * sm->m;
  int32_t* cptr = nullptr;
  uint32_t* cidx = nullptr;
  CHECK(dalloc(ctx, &cptr, N + 1, true));
  Tmp owner_ptr;   // released on an early return
  owner_ptr.p = cptr;
  if (total > 0) {
    CHECK(dalloc(ctx, &cidx, total, false));
    Tmp owner_idx;
    owner_idx.p = cidx;
    Tmp counts, iota, keys;
    CHECK(tmp_alloc(ctx, &counts, (size_t)(N + 1) * 4, true));
    LAUNCH(row_count_kernel, total, total, (const int32_t*)sm->inds, counts.as<int32_t>());
    size_t sbytes = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, sbytes, counts.as<int32_t>(), cptr, (int)(N + 1), ctx->stream));
    Tmp stmp;
    CHECK(tmp_alloc(ctx, &stmp, sbytes, false));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(stmp.p, sbytes, counts.as<int32_t>(), cptr, (int)(N + 1), ctx->stream));
    CHECK(tmp_alloc(ctx, &iota, (size_t)total * 4, false));
    CHECK(tmp_alloc(ctx, &keys, (size_t)total * 4, false));
    LAUNCH(iota_kernel, total, total, iota.as<uint32_t>());
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < N) ++bits;
    size_t rbytes = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, rbytes, (const uint32_t*)sm->inds, keys.as<uint32_t>(), iota.as<uint32_t>(), cidx,
                                              (int)total, 0, bits, ctx->stream));
    Tmp rtmp;
    CHECK(tmp_alloc(ctx, &rtmp, rbytes, false));
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(rtmp.p, rbytes, (const uint32_t*)sm->inds, keys.as<uint32_t>(), iota.as<uint32_t>(), cidx,
                                              (int)total, 0, bits, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the temporaries go out of scope
    owner_idx.p = nullptr;
  }
  owner_ptr.p = nullptr;
  sm->cover_ptr = cptr;
  sm->cover_idx = cidx;
  return AGGMG_OK;
}

// blocks [nb][m][m] (row-major, or one column-major Julia Matrix per block) -> inverses on the device;
// *first_singular = index of the first singular block or -1
int setup_invert_blocks(aggmg_ctx* ctx, int64_t nb, int m, const double* blocks_dev, int colmajor, double* inv_dev,
                        int64_t* first_singular) {
  unsigned long long* sing = nullptr;
  HIPCHK(hipMalloc((void**)&sing, sizeof(unsigned long long)));
  Tmp sing_owner;
  sing_owner.p = sing;
  const unsigned long long none = (unsigned long long)nb;
  HIPCHK(hipMemcpyAsync(sing, &none, sizeof(none), hipMemcpyHostToDevice, ctx->stream));
  Tmp work;
  switch (m) {
#define CASE(MM)                                                                              \
  case MM:                                                                                    \
    LAUNCH((block_invert_kernel<MM>), nb, nb, blocks_dev, colmajor, inv_dev, sing);           \
    break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    default:
      CHECK(tmp_alloc(ctx, &work, (size_t)nb * ((size_t)m * m + 2 * m) * sizeof(double), false));
      LAUNCH(block_invert_any_kernel, nb, nb, m, blocks_dev, colmajor, work.as<double>(), inv_dev, sing);
  }
  unsigned long long got = none;
  HIPCHK(hipMemcpyAsync(&got, sing, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *first_singular = got >= none ? -1 : (int64_t)got;
  return AGGMG_OK;
}

static bool btd_supported(int m, bool cmp) {
  if (cmp) return m >= 2 && m <= 9;
  return m >= 1 && m <= 5;
}

// dg_smoother(mesh, A, :blockJac) / cg_smoother(..., :addSchwarz | :hybridSchwarz) on the device: index lists
// -> dense blocks A[inds, inds] -> pivoted LU -> inverses; contiguous aligned lists on a block-tridiagonal
// operator additionally get the index-free fused form (patterns and symmetry detected on the device)
int setup_block_smoother(aggmg_ctx* ctx, aggmg_smoother* sm, const int64_t* blockinds, int one_based, int want_btd) {
  aggmg_op* A = sm->A;
  const int64_t N = sm->N, nb = sm->nb;
  const int m = (int)sm->m;
  const int64_t total = nb * m;
  const int64_t base = one_based ? 1 : 0;
  Tmp inds64;
  CHECK(tmp_alloc(ctx, &inds64, (size_t)std::max<int64_t>(total, 1) * 8, false));
  if (total) HIPCHK(hipMemcpyAsync(inds64.p, blockinds, (size_t)total * 8, hipMemcpyHostToDevice, ctx->stream));
  CHECK(dalloc(ctx, &sm->inds, total, false));
  CHECK(dalloc(ctx, &sm->counts, N, true));
  Flags f;
  CHECK(f.init(ctx, 4));
  LAUNCH(inds_convert_kernel, total, total, N, inds64.as<int64_t>(), base, sm->inds, sm->counts, f.d);
  int h[4];
  CHECK(f.read(ctx, h));
  if (h[0]) return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_blockjacobi_setup: block index out of range");
  sm->contiguous = (total == N) && !h[1];
  sm->overlapping = h[2] != 0;
  CHECK(dalloc(ctx, &sm->binv, total * m, false));
  int64_t sing = -1;
  auto b = std::make_shared<BtdDev>();
  bool btd_ok = false;
  if (sm->contiguous && !sm->overlapping && m <= 9) {
    // scatter the entries into the three block diagonals: the diagonal blocks ARE A[inds, inds]
    b->m = m;
    b->ne = nb;
    CHECK(dalloc(ctx, &b->dblk, N * m, true));
    CHECK(dalloc(ctx, &b->sub, N * m, true));
    CHECK(dalloc(ctx, &b->sup, N * m, true));
    Flags f2;
    CHECK(f2.init(ctx, 4));
    unsigned* masks = reinterpret_cast<unsigned*>(f2.d + 2);
    LAUNCH(btd_scatter_kernel, N, N, m, (const int32_t*)A->csc.rowptr, (const int32_t*)A->csc.colind,
           (const double*)A->csc.vals, b->dblk, b->sub, b->sup, f2.d, masks);
    int g[4];
    CHECK(f2.read(ctx, g));
    btd_ok = !g[0];
    CHECK(setup_invert_blocks(ctx, nb, m, b->dblk, 0, sm->binv, &sing));
    if (btd_ok && want_btd && sing < 0) {
      const unsigned submask = (unsigned)g[2], supmask = (unsigned)g[3];
      bool cmp = m >= 2 && __builtin_popcount(submask) <= 1 && __builtin_popcount(supmask) <= 1;
      int c_sub = submask ? __builtin_ctz(submask) : 0, r_sup = supmask ? __builtin_ctz(supmask) : 0;
      if (cmp && !btd_supported(m, true)) cmp = false;
      if (!cmp) c_sub = r_sup = 0;
      if (cmp || btd_supported(m, false)) {
        b->cmp = cmp;
        b->c_sub = c_sub;
        b->r_sup = r_sup;
        // the fused kernel's copy of the inverses
        CHECK(dalloc(ctx, &b->binv, N * m, false));
        HIPCHK(hipMemcpyAsync(b->binv, sm->binv, (size_t)N * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        Flags asym;
        CHECK(asym.init(ctx, 1));
        const bool grp = cmp ? (m == 2 || m == 4 || m == 8) : (m == 2 || m == 4);
        const bool try_sym = grp && ctx->sym_packing && (!cmp || c_sub == r_sup);
        if (cmp) {
          CHECK(dalloc(ctx, &b->scol, N, false));
          CHECK(dalloc(ctx, &b->pcol, N, false));
          CHECK(dalloc(ctx, &b->qrow, N, false));
          LAUNCH(btd_cmp_finish_kernel, nb, nb, m, c_sub, r_sup, (const double*)b->binv, (const double*)b->sub,
                 (const double*)b->sup, b->scol, b->pcol, b->qrow);
        } else {
          CHECK(dalloc(ctx, &b->P, N * m, false));
          CHECK(dalloc(ctx, &b->Q, N * m, false));
          LAUNCH(btd_dense_finish_kernel, N, nb, m, (const double*)b->binv, (const double*)b->sub, (const double*)b->sup,
                 b->P, b->Q);
        }
        if (try_sym) {
          LAUNCH(btd_sym_check_kernel, nb, nb, m, cmp ? 1 : 0, c_sub, r_sup, 1e-13, (const double*)b->binv,
                 (const double*)b->sub, (const double*)b->sup, asym.d);
          int a1 = 0;
          CHECK(asym.read(ctx, &a1));
          if (!a1) {
            CHECK(dalloc(ctx, &b->bsym, nb * (int64_t)(m * (m + 1) / 2), false));
            LAUNCH(btd_sym_pack_kernel, nb, nb, m, (const double*)b->binv, b->bsym);
          }
        }
        if (cmp) {  // the dense off-diagonal blocks are not read by the compressed kernels
          (void)hipStreamSynchronize(ctx->stream);
          (void)hipFree(b->sub);
          (void)hipFree(b->sup);
          b->sub = b->sup = nullptr;
        }
        sm->btd = b;
        A->btd = b;
      }
    }
  } else {
    Tmp blocks;
    CHECK(tmp_alloc(ctx, &blocks, (size_t)std::max<int64_t>(total * m, 1) * sizeof(double), false));
    LAUNCH(block_extract_generic_kernel, total * m, nb, m, (const int32_t*)sm->inds, (const int32_t*)A->csc.rowptr,
           (const int32_t*)A->csc.colind, (const double*)A->csc.vals, blocks.as<double>());
    CHECK(setup_invert_blocks(ctx, nb, m, blocks.as<double>(), 0, sm->binv, &sing));
  }
  if (sing >= 0)
    return fail(ctx, AGGMG_ERR_SINGULAR,
                "aggmg_blockjacobi_setup: singular block " + std::to_string(sing + 1) + " (SingularException)");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return AGGMG_OK;
}

// ---------------------------------------------------------------------------------------------
// structured transfer of a block-tridiagonal level
// ---------------------------------------------------------------------------------------------
// two-mode transfers whose first column is exactly 1.0 in every row (the constant mode of an agglomerate's modal basis
// at the fine nodes): keep the second column on its own, the fused kernels then read 8 instead of 16 bytes per row
static int setup_unit_column(aggmg_ctx* ctx, TransferBtd* t, int64_t Nf) {
  static const bool on = [] {
    const char* e = std::getenv("AGGMG_UNIT_COLUMN");
    return !(e && e[0] == '0');
  }();
  if (!on || t->mc != 2 || Nf == 0) return AGGMG_OK;
  Flags f;
  CHECK(f.init(ctx, 1));
  LAUNCH(transfer_unit_check_kernel, Nf, Nf, (const double*)t->lf, f.d);
  int notone = 0;
  CHECK(f.read(ctx, &notone));
  if (notone) return AGGMG_OK;
  CHECK(dalloc(ctx, &t->lf1, Nf, false));
  LAUNCH(transfer_second_column_kernel, Nf, Nf, (const double*)t->lf, t->lf1);
  return AGGMG_OK;
}

int setup_transfer_btd(aggmg_ctx* ctx, const aggmg_op* L, const BtdDev* Abtd, int mf, int64_t nef, int hint_mc,
                       TransferBtd* out, bool* ok) {
  *ok = false;
  const int64_t Nf = L->m, Nc = L->n;
  if (Nf != nef * mf) return AGGMG_OK;
  Flags bad;
  CHECK(bad.init(ctx, 2));  // [0]: pattern does not fit, [1]: largest agglomerate (transfer_vr_kernel)
  for (int mc = 1; mc <= 16; ++mc) {
    if (hint_mc > 0 && mc != hint_mc) continue;
    if (Nc % mc) continue;
    const int64_t nec = Nc / mc;
    if (nec == 0 || nef % nec) continue;
    const int64_t rho = nef / nec;
    if (rho > 64) continue;
    double* lf = nullptr;
    CHECK(dalloc(ctx, &lf, Nf * mc, true));
    CHECK(bad.clear(ctx));
    LAUNCH(transfer_scatter_kernel, Nc, Nc, mf, mc, rho, (const int32_t*)L->csc.rowptr, (const int32_t*)L->csc.colind,
           (const double*)L->csc.vals, lf, bad.d);
    int bf[2] = {0, 0};
    CHECK(bad.read(ctx, bf));
    const int b1 = bf[0];
    if (b1) {
      (void)hipFree(lf);
      continue;
    }
    out->lf = lf;
    if (Abtd && Abtd->dblk) {
      CHECK(dalloc(ctx, &out->ld, Nf * mc, false));
      LAUNCH(transfer_ld_kernel, Nf, nef, mf, mc, (const double*)lf, (const double*)Abtd->dblk, out->ld);
    }
    out->mc = mc;
    out->rho = (int)rho;
    out->nec = nec;
    CHECK(setup_unit_column(ctx, out, Nf));
    *ok = true;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return AGGMG_OK;
  }
  // agglomerates of different sizes (contiguous runs of fine elements, at most 64 each)
  if (nef >= ((int64_t)1 << 31)) return AGGMG_OK;
  for (int mc = 1; mc <= 16; ++mc) {
    if (hint_mc > 0 && mc != hint_mc) continue;
    if (Nc % mc) continue;
    const int64_t nec = Nc / mc;
    if (nec == 0 || nec > nef) continue;
    double* lf = nullptr;
    int32_t *first = nullptr, *parent = nullptr;
    CHECK(dalloc(ctx, &lf, Nf * mc, true));
    CHECK(dalloc(ctx, &first, nec + 1, true));
    CHECK(dalloc(ctx, &parent, nef, true));
    CHECK(bad.clear(ctx));
    LAUNCH(transfer_vr_kernel, nec, nec, nef, mf, mc, 64, (const int32_t*)L->csc.rowptr, (const int32_t*)L->csc.colind,
           (const double*)L->csc.vals, first, parent, lf, bad.d);
    int bf[2] = {0, 0};
    CHECK(bad.read(ctx, bf));
    const int b1 = bf[0];
    if (b1) {
      (void)hipFree(lf), (void)hipFree(first), (void)hipFree(parent);
      continue;
    }
    out->lf = lf;
    out->first = first;
    out->parent = parent;
    if (Abtd && Abtd->dblk) {
      CHECK(dalloc(ctx, &out->ld, Nf * mc, false));
      LAUNCH(transfer_ld_kernel, Nf, nef, mf, mc, (const double*)lf, (const double*)Abtd->dblk, out->ld);
    }
    out->mc = mc;
    out->rho = 0;
    out->maxagg = bf[1];
    out->nec = nec;
    CHECK(setup_unit_column(ctx, out, Nf));
    *ok = true;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return AGGMG_OK;
  }
  return AGGMG_OK;
}

// ---------------------------------------------------------------------------------------------
// cyclic-reduction factorisation of the coarsest operator
// ---------------------------------------------------------------------------------------------
template <int M>
static int cr_levels_t(aggmg_ctx* ctx, CrDev* cr, double* a, double* b, double* c, int64_t n, double* cond_dev, int* bad_dev) {
  const int mm2 = M * M;
  auto own = [&](void* p) { cr->owned.push_back(p); };
  while (n > 1) {
    if (M <= 2 && n <= 1024) {  // a candidate for the tail's first level (parallel cyclic reduction, setup_pcr)
      CrDev::Raw R{(int)cr->lv.size(), n, nullptr, nullptr, nullptr};
      for (double** p : {&R.a, &R.b, &R.c}) CHECK(dalloc(ctx, p, n * mm2, false));
      HIPCHK(hipMemcpyAsync(R.a, a, n * mm2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(R.b, b, n * mm2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(R.c, c, n * mm2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      cr->raw.push_back(R);
    }
    const int64_t ne = (n + 1) / 2, no = n / 2;
    double *lu = nullptr, *Za = nullptr, *Zc = nullptr, *a2 = nullptr, *b2 = nullptr, *c2 = nullptr;
    int32_t* perm = nullptr;
    CHECK(dalloc(ctx, &lu, no * mm2, false));
    own(lu);
    CHECK(dalloc(ctx, &perm, no * M, false));
    own(perm);
    Tmp tza, tzc;
    CHECK(tmp_alloc(ctx, &tza, (size_t)std::max<int64_t>(no, 1) * mm2 * sizeof(double), false));
    CHECK(tmp_alloc(ctx, &tzc, (size_t)std::max<int64_t>(no, 1) * mm2 * sizeof(double), false));
    Za = tza.as<double>();
    Zc = tzc.as<double>();
    CHECK(dalloc(ctx, &a2, ne * mm2, false));
    CHECK(dalloc(ctx, &b2, ne * mm2, false));
    CHECK(dalloc(ctx, &c2, ne * mm2, false));
    LAUNCH((cr_factor_odd_kernel<M>), no, no, (const double*)a, (const double*)b, (const double*)c, lu, perm, Za, Zc, cond_dev,
           bad_dev);
    LAUNCH((cr_schur_even_kernel<M>), ne, n, ne, (const double*)a, (const double*)b, (const double*)c, (const double*)Za,
           (const double*)Zc, a2, b2, c2);
    HIPCHK(hipStreamSynchronize(ctx->stream));  // Za / Zc leave scope
    // the solve reads the off-diagonal blocks parity-split (forward: even rows, backward: odd rows)
    double *fe = nullptr, *fo = nullptr;
    CHECK(dalloc(ctx, &fe, ne * 2 * mm2, false));
    own(fe);
    CHECK(dalloc(ctx, &fo, std::max<int64_t>(no, 1) * 2 * mm2, false));
    own(fo);
    LAUNCH(cr_split_kernel, n * 2 * mm2, n, mm2, (const double*)a, (const double*)c, fe, fo);
    LAUNCH((cr_even_multipliers_kernel<M>), no, no, ne, (const double*)lu, (const int32_t*)perm, fe);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    CrLevel L;
    L.n = n;
    L.n_even = ne;
    L.n_odd = no;
    L.fe = fe;
    L.fo = fo;
    L.lu = lu;
    L.perm = perm;
    (void)hipFree(a), (void)hipFree(c);
    (void)hipFree(b);  // the diagonal blocks of this level live on in the odd factors and the next level
    cr->lv.push_back(L);
    a = a2;
    b = b2;
    c = c2;
    n = ne;
    if ((int)cr->lv.size() > kCrMaxLevels) {
      (void)hipFree(a), (void)hipFree(b), (void)hipFree(c);
      return AGGMG_ERR_UNSUPPORTED;
    }
  }
  double* lu = nullptr;
  int32_t* perm = nullptr;
  CHECK(dalloc(ctx, &lu, mm2, false));
  own(lu);
  CHECK(dalloc(ctx, &perm, M, false));
  own(perm);
  hipLaunchKernelGGL((cr_factor_last_kernel<M>), dim3(1), dim3(64), 0, ctx->stream, (const double*)b, lu, perm, cond_dev, bad_dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  (void)hipFree(a), (void)hipFree(b), (void)hipFree(c);
  cr->lu_last = lu;
  cr->perm_last = perm;
  return AGGMG_OK;
}

static void cr_raw_release(CrDev* c) {
  for (auto& R : c->raw)
    for (double* p : {R.a, R.b, R.c})
      if (p) (void)hipFree(p);
  c->raw.clear();
}

// parallel cyclic reduction of the tail's system (internal.hpp CrDev::Pcr): log2(n) levels of multipliers from the blocks
// kept in `raw`; leaves pcr.valid false when a diagonal block of some level is singular (the register-blocked tail stays)
template <int M>
static int setup_pcr_t(aggmg_ctx* ctx, CrDev* cr, const CrDev::Raw& R) {
  const int64_t n = R.n;
  constexpr int MM = M * M;
  int L = 0;
  while (((int64_t)1 << L) < n) ++L;
  if (L < 1 || L > kPcrMaxLevels) return AGGMG_OK;
  double *mult = nullptr, *lu = nullptr, *a2 = nullptr, *b2 = nullptr, *c2 = nullptr;
  int32_t* perm = nullptr;
  CHECK(dalloc(ctx, &mult, (int64_t)L * n * 2 * MM, true));
  CHECK(dalloc(ctx, &lu, n * MM, false));
  CHECK(dalloc(ctx, &perm, n * M, false));
  Tmp ta, tb, tc;
  CHECK(tmp_alloc(ctx, &ta, (size_t)n * MM * sizeof(double), false));
  CHECK(tmp_alloc(ctx, &tb, (size_t)n * MM * sizeof(double), false));
  CHECK(tmp_alloc(ctx, &tc, (size_t)n * MM * sizeof(double), false));
  a2 = ta.as<double>(), b2 = tb.as<double>(), c2 = tc.as<double>();
  Flags bad;
  CHECK(bad.init(ctx, 1));
  double *a = R.a, *b = R.b, *c = R.c;  // (the copies are scratch from here on: ping-pong with a2, b2, c2)
  for (int k = 0; k < L; ++k) {
    LAUNCH((pcr_factor_kernel<M>), n, n, (const double*)b, lu, perm, bad.d);
    LAUNCH((pcr_reduce_kernel<M>), n, n, (int64_t)1 << k, (const double*)a, (const double*)b, (const double*)c,
           (const double*)lu, (const int32_t*)perm, mult + (int64_t)k * n * 2 * MM, a2, b2, c2);
    std::swap(a, a2), std::swap(b, b2), std::swap(c, c2);
  }
  LAUNCH((pcr_factor_kernel<M>), n, n, (const double*)b, lu, perm, bad.d);
  int b1 = 0;
  CHECK(bad.read(ctx, &b1));
  if (b1) {
    (void)hipFree(mult), (void)hipFree(lu), (void)hipFree(perm);
    return AGGMG_OK;
  }
  cr->owned.push_back(mult), cr->owned.push_back(lu), cr->owned.push_back(perm);
  cr->pcr.n = (int)n;
  cr->pcr.L = L;
  cr->pcr.mult = mult;
  cr->pcr.lu = lu;
  cr->pcr.perm = perm;
  cr->pcr.valid = true;
  return AGGMG_OK;
}

static void cr_release(CrDev* c) {
  cr_raw_release(c);
  for (void* p : c->owned)
    if (p) (void)hipFree(p);
  *c = CrDev();
}

void cr_discard(CrDev* c) { cr_release(c); }

// probe vector of the factorisation check: entries in [-1, 1) from a hash of the index (no structure a
// tridiagonal stencil could annihilate)
__global__ __launch_bounds__(kSetupThreads) void probe_vector_kernel(int64_t n, double* __restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (i >= n) return;
  uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  w[i] = (double)(int64_t)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
}

// y += sign * A x from the uploaded CSC arrays, set-up checks only.  DETERMINISTIC: the acceptance decisions taken on
// its result (probe backward error, parallel tail kept or not) must come out the same run to run and on every rank
// of a partitioned job that holds a replica of the operator -- no atomics.  The operator is block-tridiagonal with
// block size m (what the cyclic reduction it checks has established), so row i has its entries in the columns of
// blocks i/m - 1 .. i/m + 1: one thread per row walks those <= 3m columns in ascending order and looks its row up in
// each (rows ascend inside a column: validated at upload).
__global__ __launch_bounds__(kSetupThreads) void csc_band_gather_kernel(int64_t n, int m, const int32_t* __restrict__ cp,
                                                                        const int32_t* __restrict__ rv,
                                                                        const double* __restrict__ vv,
                                                                        const double* __restrict__ x, double sign,
                                                                        double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (i >= n) return;
  const int64_t bi = i / m;
  const int64_t j0 = bi > 0 ? (bi - 1) * m : 0;
  const int64_t j1 = (bi + 2) * m < n ? (bi + 2) * m : n;
  double acc = 0.0;
  for (int64_t j = j0; j < j1; ++j) {
    int32_t lo = cp[j], hi = cp[j + 1];
    while (lo < hi) {  // first entry of column j with row >= i
      const int32_t mid = lo + (hi - lo) / 2;
      if (rv[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
    }
    if (lo < cp[j + 1] && rv[lo] == (int32_t)i) acc += vv[lo] * x[j];
  }
  y[i] += sign * acc;
}

int setup_probe_vector(aggmg_ctx* ctx, int64_t n, double* w) {
  LAUNCH(probe_vector_kernel, n, n, w);
  return AGGMG_OK;
}

// a smooth positive vector, 1 + cos(pi i / n) / 2: as a RIGHT-HAND SIDE of an elliptic coarse operator it asks for a large
// smooth solution -- where a reduction that accumulates like an inverse loses most against one that substitutes back
__global__ __launch_bounds__(kSetupThreads) void smooth_vector_kernel(int64_t n, double* __restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * kSetupThreads + threadIdx.x;
  if (i >= n) return;
  w[i] = 1.0 + 0.5 * cos(3.141592653589793 * (double)i / (double)n);
}

int setup_smooth_vector(aggmg_ctx* ctx, int64_t n, double* w) {
  LAUNCH(smooth_vector_kernel, n, n, w);
  return AGGMG_OK;
}

int setup_band_matvec_add(aggmg_ctx* ctx, const aggmg_op* A, int m, const double* x, double sign, double* y) {
  LAUNCH(csc_band_gather_kernel, A->m, A->m, m, (const int32_t*)A->csc.rowptr, (const int32_t*)A->csc.colind,
         (const double*)A->csc.vals, x, sign, y);
  return AGGMG_OK;
}

// Sets cr->valid when the operator is block-tridiagonal for some block size m <= 8 and every pivot block is
// comfortably invertible; leaves it false otherwise (the caller then keeps the host banded solver).
int setup_cr(aggmg_ctx* ctx, const aggmg_op* Ac, int hint_m, CrDev* cr) {
  cr->valid = false;
  const int64_t N = Ac->m;
  if (N == 0) return AGGMG_OK;
  const int32_t* cp = Ac->csc.rowptr;
  const int32_t* rv = Ac->csc.colind;
  const double* vv = Ac->csc.vals;
  Flags f;
  CHECK(f.init(ctx, 4));
  LAUNCH(band_kernel, N, N, cp, rv, f.d);
  int band[4];
  CHECK(f.read(ctx, band));
  const int kl = band[0], ku = band[1];
  Flags bad;
  CHECK(bad.init(ctx, 1));
  auto fits = [&](int mm_, bool* out) -> int {
    CHECK(bad.clear(ctx));
    LAUNCH(block_fit_kernel, N, N, mm_, cp, rv, bad.d);
    int b1 = 0;
    CHECK(bad.read(ctx, &b1));
    *out = !b1;
    return AGGMG_OK;
  };
  int m = 0;
  bool okm = false;
  if (hint_m >= 1 && hint_m <= 8) {
    CHECK(fits(hint_m, &okm));
    if (okm) m = hint_m;
  }
  // otherwise the smallest block size for which the operator is block-tridiagonal
  for (int cand = 1; !m && cand <= 8; ++cand)
    if (std::max(kl, ku) <= 2 * cand - 1) {
      CHECK(fits(cand, &okm));
      if (okm) m = cand;
    }
  if (!m) return AGGMG_OK;
  const int mm2 = m * m;
  const int64_t n = (N + m - 1) / m;
  cr->m = m;
  cr->n0 = n;
  cr->N = N;
  double *a = nullptr, *b = nullptr, *c = nullptr;
  CHECK(dalloc(ctx, &a, n * mm2, true));
  CHECK(dalloc(ctx, &b, n * mm2, true));
  CHECK(dalloc(ctx, &c, n * mm2, true));
  LAUNCH(cr_pack_kernel, n * m, N, m, cp, rv, vv, a, b, c);
  Tmp condt;
  CHECK(tmp_alloc(ctx, &condt, sizeof(double), true));
  CHECK(bad.clear(ctx));
  int st = AGGMG_OK;
  switch (m) {
#define CASE(MM) \
  case MM:       \
    st = cr_levels_t<MM>(ctx, cr, a, b, c, n, condt.as<double>(), bad.d); \
    break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  if (st == AGGMG_ERR_UNSUPPORTED) {  // too many levels
    cr_release(cr);
    return AGGMG_OK;
  }
  if (st != AGGMG_OK) {
    cr_release(cr);
    return st;
  }
  int b1 = 0;
  CHECK(bad.read(ctx, &b1));
  double cond = 0.0;
  HIPCHK(hipMemcpyAsync(&cond, condt.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  cr->cond_est = cond;
  if (b1 || !(cond < 1e13)) {  // a pivot block is singular or close to it: keep the pivoted banded LU
    cr_release(cr);
    return AGGMG_OK;
  }
  // plan: chunk stages (one workgroup per 2^q-block chunk) until what is left fits the
  // single-workgroup tail (cr_kernels.hpp)
  const int nl = (int)cr->lv.size();
  auto level_n = [&](int l) -> int64_t { return l < nl ? cr->lv[l].n : 1; };
  auto dz = [&](int64_t len, double** out) -> int {
    CHECK(dalloc(ctx, out, len, true));
    cr->owned.push_back(*out);
    return AGGMG_OK;
  };
  // (AGGMG_CR_TAIL_ROWS / AGGMG_CR_MAX_Q shrink the tail and the chunks: the tests use them to run the
  // several-stage plan of systems beyond 2^24 rows at sizes a CPU reference solves in seconds)
  auto env_int = [](const char* name, int dflt, int lo, int hi) {
    const char* e = std::getenv(name);
    const int v = e && *e ? std::atoi(e) : dflt;
    return std::min(std::max(v, lo), hi);
  };
  const int tail_rows = env_int("AGGMG_CR_TAIL_ROWS", kCrTailRows, 8, kCrTailRows);
  const int max_q = env_int("AGGMG_CR_MAX_Q", ctx->cr_max_q, 1, kCrMaxStageLevels);
  CrSolvePlan plan;
  {
    const char* e = std::getenv("AGGMG_CR_FILL");
    const int fmax = e && *e ? std::atoi(e) : 12;
    const char* w = std::getenv("AGGMG_CR_MINWG");
    const int minwg = w && *w ? std::atoi(w) : 256;
    std::vector<int64_t> ln(nl);
    for (int l = 0; l < nl; ++l) ln[l] = cr->lv[l].n;
    // (large chunks -- every thread at least one sub-chunk of the streaming first step -- as long as a few hundred
    // workgroups remain; host_plan.hpp)
    if (!cr_plan_solve(ln, m, tail_rows, max_q, fmax, minwg, &plan)) {
      cr_release(cr);
      return AGGMG_OK;
    }
  }
  for (const CrStagePlan& P : plan.stages) {
    CrStage S;
    static_cast<CrStagePlan&>(S) = P;
    const int64_t nb = (S.n_out * m + 31) & ~(int64_t)31;  // the three boundary vectors in one allocation
    CHECK(dz(3 * nb, &S.partR));
    S.partL = S.partR + nb;  // partL[0] is never written: stays zero
    S.xq = S.partL + nb;
    if (S.stack_stride > 0) CHECK(dz((S.n_out + 1) * (int64_t)S.stack_stride, &S.stack));
    if (S.mid_total > 0) CHECK(dz(S.mid_total, &S.mid));
    cr->st.push_back(S);
  }
  static_cast<CrStagePlan&>(cr->tail) = plan.tail;
  if (cr->tail.mid_total > 0) CHECK(dz(cr->tail.mid_total, &cr->tail.mid));
  // The small levels (the tail and the last steps of the stage before it) are worked through by a
  // few threads, one dependent step after the other: their factors go into ONE allocation, so that a
  // step touches a couple of pages instead of four arrays per level each on a page of its own -- an
  // address-translation miss per array is what such a step would otherwise wait for.
  {
    int pf = cr->tail.l0;
    while (pf > 0 && level_n(pf - 1) * m <= 8 * kCrTailRows) --pf;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t mm = (size_t)m * m;
    size_t total = al(mm * sizeof(double)) + al(m * sizeof(int32_t));
    for (int l = pf; l < nl; ++l) {
      const CrLevel& L = cr->lv[l];
      const size_t no = (size_t)std::max<int64_t>(L.n_odd, 1);
      total += al(L.n_even * 2 * mm * sizeof(double)) + al(no * 2 * mm * sizeof(double)) + al(no * mm * sizeof(double)) +
               al(no * m * sizeof(int32_t));
    }
    char* arena = nullptr;
    HIPCHK(hipMalloc((void**)&arena, total));
    size_t off = 0;
    std::vector<void*> old;
    auto move = [&](const void* src, size_t bytes) -> void* {
      void* dst = arena + off;
      (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream);
      off += al(bytes);
      old.push_back(const_cast<void*>(src));
      return dst;
    };
    for (int l = pf; l < nl; ++l) {
      CrLevel& L = cr->lv[l];
      const size_t no = (size_t)std::max<int64_t>(L.n_odd, 1);
      L.fe = (const double*)move(L.fe, L.n_even * 2 * mm * sizeof(double));
      L.fo = (const double*)move(L.fo, no * 2 * mm * sizeof(double));
      L.lu = (const double*)move(L.lu, no * mm * sizeof(double));
      L.perm = (const int32_t*)move(L.perm, no * m * sizeof(int32_t));
    }
    cr->lu_last = (const double*)move(cr->lu_last, mm * sizeof(double));
    cr->perm_last = (const int32_t*)move(cr->perm_last, m * sizeof(int32_t));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (void* q : old) {
      auto it = std::find(cr->owned.begin(), cr->owned.end(), q);
      if (it != cr->owned.end()) cr->owned.erase(it);
      (void)hipFree(q);
    }
    cr->owned.push_back(arena);
  }
  // the tail's system by parallel cyclic reduction where it applies (AGGMG_CR_PCR=0: off)
  if (m <= 2 && cr->tail.nsteps >= 1 && env_int("AGGMG_CR_PCR", 1, 0, 1)) {
    // up to 512 rows: all of them in the parallel part; 513 .. 1024: its even rows (the next level's system)
    const bool pre = cr->tail.n_in > 512;
    const int want_level = cr->tail.l0 + (pre ? 1 : 0);
    const int64_t want_n = pre ? (cr->tail.n_in + 1) / 2 : cr->tail.n_in;
    if (cr->tail.n_in <= 1024 && (!pre || cr->tail.q >= 2))
      for (const auto& R : cr->raw)
        if (R.level == want_level && R.n == want_n) {
          if (m == 1) CHECK(setup_pcr_t<1>(ctx, cr, R));
          if (m == 2) CHECK(setup_pcr_t<2>(ctx, cr, R));
          cr->pcr.pre = pre;
        }
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  cr_raw_release(cr);
  if (n * m != N) {
    CHECK(dz(n * m, &cr->d0));
    CHECK(dz(n * m, &cr->x0));
  }
  {
    void* t = nullptr;
    HIPCHK(hipMalloc(&t, sizeof(unsigned int)));
    cr->owned.push_back(t);
    cr->ticket = (unsigned int*)t;
    HIPCHK(hipMemsetAsync(t, 0, sizeof(unsigned int), ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  cr->valid = true;
  return AGGMG_OK;
}

// ---------------------------------------------------------------------------------------------
// CG chain form
// ---------------------------------------------------------------------------------------------
static int cgt_build_impl(aggmg_ctx* ctx, aggmg_smoother* sm, const int64_t* elems, int64_t m1, int64_t nel, int one_based,
                          bool generate);

int cgt_build(aggmg_ctx* ctx, aggmg_smoother* sm, const int64_t* elems, int64_t m1, int64_t nel, int one_based) {
  return cgt_build_impl(ctx, sm, elems, m1, nel, one_based, false);
}

// AGGMG_OPT_DETECT_CHAIN: no element lists came with the operator -- try the lists of a CgMesh of degree p on
// (N - 1) / p elements in the reference's vertices-first numbering, generated on the device, for every p that divides
// N - 1 and whose entry count is plausible; cgt_build's own checks (a chain, every coupling inside the pattern) decide
int cgt_detect(aggmg_ctx* ctx, aggmg_smoother* sm) {
  const aggmg_op* A = sm->A;
  const int64_t N = A->m;
  if (N < 3 || A->m != A->n) return AGGMG_OK;
  for (int64_t p = 8; p >= 1 && !sm->cgt; --p) {
    if ((N - 1) % p) continue;
    const int64_t nel = (N - 1) / p;
    // a CG operator of degree p holds at most nel (p + 1)^2 entries, and not much fewer (Dirichlet rows are emptied)
    const double full = (double)nel * (double)((p + 1) * (p + 1));
    if ((double)A->nnz > full || (double)A->nnz < 0.5 * full) continue;
    CHECK(cgt_build_impl(ctx, sm, nullptr, p + 1, nel, 0, true));
  }
  return AGGMG_OK;
}

static int cgt_build_impl(aggmg_ctx* ctx, aggmg_smoother* sm, const int64_t* elems, int64_t m1, int64_t nel, int one_based,
                          bool generate) {
  aggmg_op* A = sm->A;
  const int64_t N = A->m;
  const int64_t p = m1 - 1;
  if (p < 1 || p > 8 || nel < 1) return AGGMG_OK;  // generic path
  if (N != nel * p + 1) return AGGMG_OK;
  const int64_t base = one_based ? 1 : 0;
  const int m = (int)p;
  const int64_t ne = nel + 1, Np = ne * m;
  if (Np >= ((int64_t)1 << 31)) return AGGMG_OK;
  auto g = std::make_shared<CgtDev>();
  g->m = m;
  g->ne = ne;
  g->N = N;
  Tmp el64;
  CHECK(tmp_alloc(ctx, &el64, (size_t)nel * m1 * 8, false));
  if (generate) {
    LAUNCH(chain_generate_elements_kernel, nel, nel, m, el64.as<int64_t>());
  } else {
    HIPCHK(hipMemcpyAsync(el64.p, elems, (size_t)nel * m1 * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  CHECK(dalloc(ctx, &g->perm, Np, false));
  CHECK(dalloc(ctx, &g->inv, N, false));
  HIPCHK(hipMemsetAsync(g->perm, 0xFF, (size_t)Np * 4, ctx->stream));
  HIPCHK(hipMemsetAsync(g->inv, 0xFF, (size_t)N * 4, ctx->stream));
  Flags f;
  CHECK(f.init(ctx, 4));
  LAUNCH(chain_perm_kernel, nel, nel, m, N, el64.as<int64_t>(), base, g->perm, f.d);
  LAUNCH(perm_invert_kernel, Np, Np, (const int32_t*)g->perm, g->inv, f.d);
  LAUNCH(perm_cover_kernel, N, N, (const int32_t*)g->inv, f.d);
  LAUNCH(chain_affine_check_kernel, Np, ne, m, (const int32_t*)g->perm, f.d);
  int h[4];
  CHECK(f.read(ctx, h));
  if (h[0]) return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_jacobi_setup_elements: node index out of range");
  if (h[1]) return AGGMG_OK;  // not a chain: generic path
  g->affine = !h[3];
  CHECK(dalloc(ctx, &g->dblk, Np * m, true));
  CHECK(dalloc(ctx, &g->subrow, Np, true));
  CHECK(dalloc(ctx, &g->supcol, Np, true));
  LAUNCH(chain_scatter_kernel, N, N, m, (const int32_t*)g->inv, (const int32_t*)A->csc.rowptr, (const int32_t*)A->csc.colind,
         (const double*)A->csc.vals, g->dblk, g->subrow, g->supcol, f.d);
  LAUNCH(chain_pad_kernel, Np, Np, m, (const int32_t*)g->perm, g->dblk);
  CHECK(f.read(ctx, h));
  if (h[2]) return AGGMG_OK;  // couplings beyond the chain pattern: generic path
  sm->cgt = g;
  A->cgt = g;
  return AGGMG_OK;
}

// Element Schwarz smoothers on a chain (cg_smoother(cgMesh, A, :addSchwarz / :hybridSchwarz), src/smoother.jl:104-134):
// after cgt_build recognised the element chain, the inverses of the element blocks (already computed for the
// generic apply, K6) are re-ordered into the rows the fused kernel keeps in registers.  sw: 1 additive, 2 hybrid.
int cgt_attach_schwarz(aggmg_ctx* ctx, aggmg_smoother* sm, int sw) {
  if (!sm->cgt || !sm->binv) return AGGMG_OK;
  CgtDev& g = *sm->cgt;
  const int M = g.m;
  const int64_t nel = g.ne - 1;
  if (sm->m != M + 1 || sm->nb != nel) return AGGMG_OK;
  CHECK(dalloc(ctx, &g.zrows, g.ne * M * (M + 1), true));
  CHECK(dalloc(ctx, &g.zlast, g.ne * (M + 1), true));
  LAUNCH(chain_schwarz_rows_kernel, nel * (M + 1), nel, M, (const double*)sm->binv, g.zrows, g.zlast);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  g.sw = sw;
  return AGGMG_OK;
}

int cgt_build_transfer(aggmg_ctx* ctx, const aggmg_op* L, const CgtDev& f, const CgtDev* coarse, int hint_mc,
                       TransferCgt* out, bool* ok) {
  const int tile_blocks = cgt_tile_blocks(f.m);
  *ok = false;
  const int64_t Nf = L->m, Nc = L->n;
  if (Nf != f.N) return AGGMG_OK;
  const int M = f.m;
  const int64_t nel = f.ne - 1, Np = f.ne * M;
  const int32_t* cp = L->csc.rowptr;
  const int32_t* rv = L->csc.colind;
  const double* vv = L->csc.vals;
  Flags fl;
  CHECK(fl.init(ctx, 4));
  int h[4];

  // ---- chain: the coarse level is a CG level on the same elements ------------------------------
  if (nel > 0 && Nc > 1 && (Nc - 1) % nel == 0 && (Nc - 1) / nel <= 8 && M >= 2) {
    const int mc = (int)((Nc - 1) / nel);
    int32_t *cperm = nullptr, *cinv = nullptr;
    Tmp own_perm, own_inv;
    bool have = false;
    if (coarse && coarse->N == Nc && coarse->m == mc && coarse->ne == f.ne) {
      cperm = coarse->perm;
      cinv = coarse->inv;
      have = true;
    } else if (!coarse) {
      // the coarse level carries no element lists (the coarsest level has no smoother): read its chain
      // off L -- a coarse column holding a fine vertex row is that block's vertex, the others are the
      // interior nodes of the one element whose fine rows they touch, in ascending order
      CHECK(tmp_alloc(ctx, &own_perm, (size_t)f.ne * mc * 4, false));
      CHECK(tmp_alloc(ctx, &own_inv, (size_t)Nc * 4, false));
      cperm = own_perm.as<int32_t>();
      cinv = own_inv.as<int32_t>();
      HIPCHK(hipMemsetAsync(cperm, 0xFF, (size_t)f.ne * mc * 4, ctx->stream));
      HIPCHK(hipMemsetAsync(cinv, 0xFF, (size_t)Nc * 4, ctx->stream));
      Tmp slots;
      CHECK(tmp_alloc(ctx, &slots, (size_t)f.ne * 4, true));
      LAUNCH(chain_coarse_vertices_kernel, Nc, Nc, M, mc, (const int32_t*)f.inv, cp, rv, cperm, cinv, fl.d);
      LAUNCH(chain_coarse_interiors_kernel, Nc, Nc, M, mc, (const int32_t*)f.inv, cp, rv, cperm, cinv, slots.as<int32_t>(), fl.d);
      LAUNCH(chain_coarse_sort_kernel, f.ne, f.ne, mc, cperm, cinv, (const int32_t*)slots.as<int32_t>(), nel, fl.d);
      CHECK(fl.read(ctx, h));
      have = !h[1];
      CHECK(fl.clear(ctx));
    }
    if (have) {
      const int w = mc + 1;
      double* l = nullptr;
      CHECK(dalloc(ctx, &l, Np * w, true));
      LAUNCH(chain_transfer_scatter_kernel, Nc, Nc, M, mc, (const int32_t*)f.inv, (const int32_t*)cinv, cp, rv, vv, l, fl.d);
      CHECK(fl.read(ctx, h));
      if (!h[0]) {
        out->l = l;
        CHECK(dalloc(ctx, &out->cperm, f.ne * mc, false));
        HIPCHK(hipMemcpyAsync(out->cperm, cperm, (size_t)f.ne * mc * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        out->type = kTrChain;
        out->mc = mc;
        out->rho = 1;
        out->nec = f.ne;
        *ok = true;
        return AGGMG_OK;
      }
      (void)hipFree(l);
      CHECK(fl.clear(ctx));
    }
  }

  // ---- agglomerating: the coarse level has contiguous blocks of mc DoFs per rho fine elements ---
  for (int mc = 1; mc <= 16; ++mc) {
    if (hint_mc > 0 && mc != hint_mc) continue;
    if (Nc % mc) continue;
    const int64_t nec = Nc / mc;
    if (nec == 0 || nel % nec) continue;
    const int64_t rho = nel / nec;
    if (rho > 64 || rho * 4 > tile_blocks) continue;
    double *l = nullptr, *lp = nullptr;
    CHECK(dalloc(ctx, &l, Np * mc, true));
    CHECK(dalloc(ctx, &lp, f.ne * mc, true));
    CHECK(fl.clear(ctx));
    LAUNCH(agg_transfer_scatter_kernel, Nc, Nc, M, mc, rho, (const int32_t*)f.inv, cp, rv, vv, l, lp, fl.d);
    CHECK(fl.read(ctx, h));
    if (h[0]) {
      (void)hipFree(l);
      (void)hipFree(lp);
      continue;
    }
    out->l = l;
    out->lp = lp;
    out->type = kTrAgg;
    out->mc = mc;
    out->rho = (int)rho;
    out->nec = nec;
    *ok = true;
    return AGGMG_OK;
  }
  return AGGMG_OK;
}
