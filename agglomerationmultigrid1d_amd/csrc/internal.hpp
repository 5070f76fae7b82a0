// Shared internals of libaggmg_hip: the objects behind the opaque handles of include/aggmg_hip.h
// and the small host helpers every translation unit of the library uses (error reporting,
// uploads, scratch vectors, the HIP-event profiler, the set-up thread pool).
#pragma once
#include "../../include/aggmg_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "cr_kernels.hpp"
#include "cgt_kernels.hpp"

using namespace aggmg;

// ---------------------------------------------------------------------------------------------
// objects behind the opaque handles
// ---------------------------------------------------------------------------------------------
inline thread_local std::string g_create_error;

struct ProfEvent {
  hipEvent_t a, b;
  int tag;
};

struct aggmg_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  bool sym_packing = true;  // AGGMG_OPT_SYMMETRIC_PACKING
  int cr_max_q = 12;        // AGGMG_OPT_COARSE_CHUNK_LOG2
  bool detect_chain = true; // AGGMG_OPT_DETECT_CHAIN
  bool mg_checkpoint = [] { const char* e = std::getenv("AGGMG_MG_CHECKPOINT"); return !(e && e[0] == '0'); }();  // AGGMG_OPT_MG_CHECKPOINT
  bool pair_levels = [] { const char* e = std::getenv("AGGMG_PAIR"); return !(e && e[0] == '0'); }();  // AGGMG_OPT_PAIR_LEVELS
  int profiling = 0;  // 0 off, 1 every launch, 2 only the fine-level fused-down launch (dominant kernel)
  std::vector<ProfEvent> prof;
  std::vector<hipEvent_t> ev_pool;
  // scratch vectors for ping-pong / temporaries, grown on demand
  double* scratch[3] = {nullptr, nullptr, nullptr};
  int64_t scratch_len[3] = {0, 0, 0};
  // outer-solver work space (aggmg_multigrid_dev, aggmg_pcg_dev, ...): vectors, dot-product
  // partials and the device-resident scalars
  double* solv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int64_t solv_len[5] = {0, 0, 0, 0, 0};
  double* solv_part = nullptr;
  double* solv_sc = nullptr;
  // host <-> device staging of the host-pointer entry points (aggmg_vcycle): per worker thread a stream and two
  // pinned chunks (HostStager in aggmg_hip.hip); allocated on first use
  struct StageLane {
    hipStream_t stream = nullptr;
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
  };
  std::vector<StageLane> stage;
  bool stage_failed = false;
  // host ranges the caller page-locked for good (aggmg_host_register / aggmg_host_alloc): copies from / to them go
  // straight over PCIe on the compute stream, no staging
  struct Pinned {
    char* base;
    size_t bytes;
    bool owned;   // allocated by aggmg_host_alloc (hipHostMalloc) rather than registered
  };
  std::vector<Pinned> pinned;  // the lanes could not be allocated once: aggmg_vcycle keeps to plain hipMemcpy
};

struct CsrDev {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  int32_t* rowptr = nullptr;
  int32_t* colind = nullptr;
  double* vals = nullptr;
  int lpr = 1;
  int32_t* rowblk = nullptr;  // CSR-stream row blocks (short-row matrices), nblk + 1 entries
  int64_t nblk = 0;
  int32_t* bandblk = nullptr; // CSR-band row blocks (square, entries within bw of the diagonal), nbandblk + 1 entries
  int64_t nbandblk = 0;
  int band_sweeps = 1;        // most point-Jacobi sweeps a csr_band_kernel launch takes on this operator (the blocks' halo is cut for it)
  int bw = -1;                // band half-width, -1: not banded / not examined
  int maxrow = -1;            // entries of the longest row (-1: not examined); <= kRowThreadMax: csr_rowthread_kernel
  CsrView view() const { return CsrView{rowptr, colind, vals, nrows}; }
};

// host copy of a CSR (only the host banded LU fallback of the coarsest solve reads an operator back)
struct HostCsr {
  std::vector<int32_t> rowptr, colind;
  std::vector<double> vals;
};

// index-free block-tridiagonal form of an operator + its block-Jacobi smoother; shared by the
// smoother that built it and the operator it describes (so aggmg_residual can use it too)
struct BtdDev {
  int m = 0;
  int64_t ne = 0;
  bool cmp = false;
  int c_sub = 0, r_sup = 0;
  double *binv = nullptr, *dblk = nullptr, *scol = nullptr, *pcol = nullptr, *qrow = nullptr;
  double* bsym = nullptr;  // packed symmetric inverses (replaces binv + pcol in the kernels) or null
  double *sub = nullptr, *sup = nullptr, *P = nullptr, *Q = nullptr;
  ~BtdDev() {
    for (double* p : {binv, dblk, scol, pcol, qrow, bsym, sub, sup, P, Q})
      if (p) (void)hipFree(p);
  }
};

// element-contiguous ("chain") form of a CG operator + its point-Jacobi smoother (cgt_kernels.hpp);
// shared by the smoother that built it and the operator it describes
struct CgtDev {
  int m = 0;        // rows per block = p
  int64_t ne = 0;   // blocks = elements + 1 (the trailing one holds the last vertex, identity-padded)
  int64_t N = 0;    // DoFs of the operator; the block-ordered vectors have ne * m entries
  double *dblk = nullptr, *subrow = nullptr, *supcol = nullptr;
  int32_t* perm = nullptr;  // [ne*m] block order -> reference numbering, -1 = padding
  int32_t* inv = nullptr;   // [N]    reference numbering -> block order
  bool affine = false;      // perm is the reference's vertices-first numbering: the kernel computes it
  // element Schwarz smoothers on the chain (cg_smoother :addSchwarz / :hybridSchwarz): rows of the inverses of
  // the element blocks A[nodes_e, nodes_e] in chain-local order [block e, first row of block e + 1]
  int sw = 0;               // 0 point Jacobi, 1 additive Schwarz, 2 hybrid Schwarz
  double* zrows = nullptr;  // [ne*m][m+1]  row i of element e's inverse at (e*m + i)
  double* zlast = nullptr;  // [ne][m+1]    its last row (the right vertex) at e + 1: with the block that owns the vertex
  ~CgtDev() {
    for (void* p : {(void*)dblk, (void*)subrow, (void*)supcol, (void*)perm, (void*)inv, (void*)zrows, (void*)zlast})
      if (p) (void)hipFree(p);
  }
};

struct aggmg_op {
  int64_t m = 0, n = 0, nnz = 0;
  int kind = AGGMG_OP_STIFFNESS;
  CsrDev csc;  // the uploaded CSC arrays (int32, 0-based) == row-gather CSR of the TRANSPOSE; always present
  CsrDev csr;  // row-gather CSR of the matrix: transposed on the device on first use (generic kernels only)
  std::shared_ptr<BtdDev> btd;  // set when a block-Jacobi smoother recognised the structure
  std::shared_ptr<CgtDev> cgt;  // set when a point-Jacobi smoother was given the CG element chain
  ~aggmg_op() {
    for (CsrDev* d : {&csc, &csr})
      for (void* p : {(void*)d->rowptr, (void*)d->colind, (void*)d->vals, (void*)d->rowblk, (void*)d->bandblk})
        if (p) (void)hipFree(p);
  }
};

struct aggmg_smoother {
  int kind = 0;  // 0 point Jacobi, 1 block (Jacobi / additive Schwarz), 2 hybrid Schwarz
  aggmg_op* A = nullptr;
  int64_t N = 0, m = 0, nb = 0;
  double* diag = nullptr;      // point Jacobi
  double* binv = nullptr;      // [nb][m][m] row-major
  int32_t* inds = nullptr;     // [nb][m]
  double* counts = nullptr;    // hybrid Schwarz
  bool overlapping = false;
  bool contiguous = false;
  // which (block, local row) entries cover each row, flat index block * m + i ascending inside a row: built on first use
  // by the generic one-pass sweep (block_sweep_kernel + block_combine_kernel)
  int32_t* cover_ptr = nullptr;   // [N + 1]
  uint32_t* cover_idx = nullptr;  // [nb * m]
  bool ordered = false;           // the blocks have been put in ascending order of their smallest index (one-pass sweep)
  bool gs = false;  // red-black block Gauss-Seidel (extension): needs the structured form
  std::shared_ptr<BtdDev> btd;  // structured fused form, or null
  std::shared_ptr<CgtDev> cgt;  // CG chain form (point Jacobi with the element lists), or null
  // owns its device arrays: every early return of a set-up routine releases what was uploaded
  ~aggmg_smoother() {
    for (void* p : {(void*)diag, (void*)binv, (void*)inds, (void*)counts, (void*)cover_ptr, (void*)cover_idx})
      if (p) (void)hipFree(p);
  }
};

struct TransferBtd {
  int mc = 0, rho = 0;   // rho: fine elements per coarse element; 0 = agglomerates of different sizes (parent / first)
  int64_t nec = 0;
  double* lf = nullptr;  // [N_f][mc]  rows of L
  double* lf1 = nullptr; // [N_f]      their second entries when mc == 2 and every first entry is exactly 1.0, else null
  double* ld = nullptr;  // [N_f][mc]  rows of (L_e' D_e)': restriction of the preconditioned residual
  int32_t* parent = nullptr;  // [ne_f]      coarse element of every fine element      (rho == 0)
  int32_t* first = nullptr;   // [ne_c + 1]  first fine element of every coarse element (rho == 0)
  int maxagg = 0;             // fine elements of the largest agglomerate                 (rho == 0)
  ~TransferBtd() {
    if (lf) (void)hipFree(lf);
    if (lf1) (void)hipFree(lf1);
    if (ld) (void)hipFree(ld);
    if (parent) (void)hipFree(parent);
    if (first) (void)hipFree(first);
  }
};

// structured transfer of a CG chain level (CgtXfer in cgt_kernels.hpp)
struct TransferCgt {
  int type = 0, mc = 0, rho = 1;
  int64_t nec = 0;
  double *l = nullptr, *lp = nullptr;
  int32_t* cperm = nullptr;  // chain: coarse block order -> coarse reference numbering
  ~TransferCgt() {
    for (void* p : {(void*)l, (void*)lp, (void*)cperm})
      if (p) (void)hipFree(p);
  }
};

struct Level {
  aggmg_op* A = nullptr;
  aggmg_smoother* S = nullptr;
  aggmg_op* L = nullptr;  // level k+1 -> k
  int64_t N = 0;
  double *u[2] = {nullptr, nullptr}, *rhs = nullptr, *tmp = nullptr;
  std::unique_ptr<TransferBtd> tb;  // structured transfer to level k+1, or null
  std::unique_ptr<TransferCgt> tc;  // CG chain level: structured transfer to level k+1, or null
  bool cgt_fused = false;           // the level runs cgt_fused_kernel (chain form + structured transfer)
  bool native_io = false;           // rhs and u[1] are kept in block order (the finer level is a fused chain level)
  int64_t Nalloc = 0;               // length of the level's vectors (ne * m for chain levels)
};

struct BandedLU {
  int64_t n = 0;
  int kl = 0, ku = 0, ldab = 0;
  std::vector<double> ab;
  std::vector<int32_t> ipiv;
};

// block cyclic reduction of the coarsest operator, factored once (device-resident)
// one launch of the cyclic reduction: levels [l0, l0 + q) in steps of up to three thread-local levels
// (cr_kernels.hpp); chunk stages reduce 2^q-block chunks to their end blocks, the tail takes the rest
struct CrStage : CrStagePlan {    // the host-side plan (host_plan.hpp) + the stage's device buffers
  double *partR = nullptr, *partL = nullptr, *xq = nullptr;  // the stage's boundary system (n_out blocks)
  double* stack = nullptr;      // per chunk: summed inputs of the steps after the first
  double* mid = nullptr;        // per step and sub-chunk: reduced right-hand sides of the inner sub-levels' odd rows
};

struct CrDev {
  bool valid = false;
  int m = 0;
  int64_t n0 = 0, N = 0;
  std::vector<CrLevel> lv;      // all reducing levels (device pointers)
  std::vector<void*> owned;     // every device allocation, for free
  const double* lu_last = nullptr;
  const int32_t* perm_last = nullptr;
  std::vector<CrStage> st;      // chunk stages ...
  CrStage tail;                 // ... then the remaining levels in one workgroup
  double *d0 = nullptr, *x0 = nullptr;  // staging for padded systems (N not a multiple of m) / in-place calls
  unsigned int* ticket = nullptr;       // last-arriving-workgroup counter of the fused forward + tail launch
  double cond_est = 0.0;
  // the tail's system by parallel cyclic reduction (cr_pcr_tail_kernel; block sizes 1, 2, up to 1024 blocks -- above 512
  // the parallel part takes the even rows, one ordinary reduction level around it): multipliers
  // of every (level, row), final diagonal blocks factored; allocations in `owned`
  struct Pcr {
    bool valid = false;
    bool pre = false;   // one ordinary cyclic-reduction level of the tail's first level around the parallel part
    int n = 0, L = 0;
    double* mult = nullptr;
    double* lu = nullptr;
    int32_t* perm = nullptr;
  } pcr;
  // set-up only: copies of the (a, b, c) blocks of the small levels, until the plan says which one the tail starts at
  struct Raw {
    int level;
    int64_t n;
    double *a, *b, *c;
  };
  std::vector<Raw> raw;
};

struct aggmg_hier {
  std::vector<Level> lv;
  int coarse_mode = 0;
  BandedLU coarse;
  CrDev cr;
  double* cyc[2] = {nullptr, nullptr};  // iterate ping-pong for multi-cycle calls (lazy)
  double* io[3] = {nullptr, nullptr, nullptr};  // x0, b, x_out of the host-pointer entry aggmg_vcycle (lazy)
  int restriction = 0;  // AGGMG_RESTRICT_EXPLICIT (default) / AGGMG_RESTRICT_PRECONDITIONED
  std::vector<double> h_coarse;
  double last_coarse_ms = 0.0;
  double cr_probe_backward_error = -1.0;  // ||d - A CR(d)|| / ||d|| of the set-up probe (-1: no device factorisation tried)
  // owns every device allocation of its levels: an early return of aggmg_hier_create releases them
  ~aggmg_hier() {
    for (auto& l : lv)
      for (double* p : {l.u[0], l.u[1], l.rhs, l.tmp})
        if (p) (void)hipFree(p);
    for (void* p : cr.owned)
      if (p) (void)hipFree(p);
    for (double* p : cyc)
      if (p) (void)hipFree(p);
    for (double* p : io)
      if (p) (void)hipFree(p);
  }
};

// The restricted residual L'(b - A u) is formed from r = b - A u evaluated with the operator's own
// entries (the reference's arithmetic, src/solvers.jl:36) -- AGGMG_RESTRICT_EXPLICIT, the default of
// every new hierarchy.  AGGMG_RESTRICT_PRECONDITIONED takes it from the sweeps' preconditioned
// residual instead, (L'D) w with w = g - P u- - Q u+ - u, which reads neither the diagonal blocks nor
// L: equal in exact arithmetic, but w inherits the rounding of the stored block inverses.  On the
// smoothest mode of the model problem that error grows like n^2 and at 2^24 fine elements turns the
// cycle from damping (x0.5, as in reference-order arithmetic) into amplifying (x2.1): measured,
// include/aggmg_hip.h and DESIGN.md section 5 -- aggmg_hier_set_restriction therefore refuses it above
// AGGMG_RESTRICT_PRECONDITIONED_MAX_ELEMS fine elements.
inline int default_restriction() { return AGGMG_RESTRICT_EXPLICIT; }

// ---------------------------------------------------------------------------------------------
// error helpers
// ---------------------------------------------------------------------------------------------
inline int fail(aggmg_ctx* ctx, int code, const std::string& msg) {
  if (ctx)
    ctx->err = msg;
  else
    g_create_error = msg;
  return code;
}

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return fail(ctx, AGGMG_ERR_HIP,                                                       \
                  std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                      std::to_string(__LINE__) + ")");                                      \
  } while (0)

#define CHECK(expr)             \
  do {                          \
    int _s = (expr);            \
    if (_s != AGGMG_OK) return _s; \
  } while (0)

template <typename T>
inline int dev_upload(aggmg_ctx* ctx, const std::vector<T>& h, T** d) {
  *d = nullptr;
  size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
  HIPCHK(hipMalloc((void**)d, bytes));
  if (!h.empty())
    HIPCHK(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return AGGMG_OK;
}

inline int scratch(aggmg_ctx* ctx, int slot, int64_t len, double** out) {
  if (ctx->scratch_len[slot] < len) {
    if (ctx->scratch[slot]) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipFree(ctx->scratch[slot]));
    }
    ctx->scratch[slot] = nullptr;
    ctx->scratch_len[slot] = 0;
    HIPCHK(hipMalloc((void**)&ctx->scratch[slot], (size_t)std::max<int64_t>(len, 1) * sizeof(double)));
    ctx->scratch_len[slot] = len;
  }
  *out = ctx->scratch[slot];
  return AGGMG_OK;
}

// ---------------------------------------------------------------------------------------------
// profiler (HIP events on the launch stream)
// ---------------------------------------------------------------------------------------------
struct ProfScope {
  aggmg_ctx* ctx;
  int idx = -1;
  ProfScope(aggmg_ctx* c, int kind, int level) : ctx(c) {
    if (!ctx->profiling) return;
    if (ctx->profiling == 2 && !(kind == AGGMG_KIND_FUSED_DOWN && level == 0)) return;
    ProfEvent pe;
    for (hipEvent_t* e : {&pe.a, &pe.b}) {
      if (!ctx->ev_pool.empty()) {
        *e = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
      } else if (hipEventCreateWithFlags(e, hipEventDisableSystemFence) != hipSuccess) {
        return;
      }
    }
    pe.tag = kind * 16 + (level & 15);
    (void)hipEventRecord(pe.a, ctx->stream);
    ctx->prof.push_back(pe);
    idx = (int)ctx->prof.size() - 1;
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(ctx->prof[idx].b, ctx->stream);
  }
};


// ---------------------------------------------------------------------------------------------
// CG chain path (cgt.hip)
// ---------------------------------------------------------------------------------------------
int cgt_tile_blocks(int m);
// checkpoints of a chain launch (CgtArgs::chk_*): in -- after `sweep`, `sweep + stride`, ... sweeps and (`final`) after the
// last one, norms' partial sums to `part`, error against `exact` (caller's numbering, may be null); out -- the launch's
// tile count (the stride of `part` between checkpoints)
struct CgtChk {
  int sweep = 0, stride = 1 << 30, final = 0;
  const double* exact = nullptr;
  double* part = nullptr;
  int64_t ntiles = 0;
};
int cgt_max_fused_sweeps(const CgtDev& g);   // sweeps one launch takes (point-Jacobi: + a residual or a closing checkpoint)
int cgt_build(aggmg_ctx* ctx, aggmg_smoother* sm, const int64_t* elems, int64_t m1, int64_t nel, int one_based);
int cgt_attach_schwarz(aggmg_ctx* ctx, aggmg_smoother* sm, int sw);
int cgt_build_transfer(aggmg_ctx* ctx, const aggmg_op* L, const CgtDev& fine, const CgtDev* coarse, int hint_mc,
                       TransferCgt* out, bool* ok);
// nsweeps sweeps on external (reference-numbered) vectors; u_in may be nullptr (zero), u_out != u_in
int cgt_smooth_ext(aggmg_ctx* ctx, const CgtDev& g, const double* u_in, const double* b, double alpha, int nsweeps,
                   double* u_out, int level, CgtChk* chk = nullptr);
int cgt_residual_ext(aggmg_ctx* ctx, const CgtDev& g, const double* u, const double* b, double* r_out);
int cgt_down(aggmg_ctx* ctx, aggmg_hier* h, int k, const double* uin, const double* rhs, int nPre, double alpha);
int cgt_up(aggmg_ctx* ctx, aggmg_hier* h, int k, const double* rhs, int nPost, double alpha, double* dst,
           const double* src = nullptr, CgtChk* chk = nullptr);
int cgt_mid(aggmg_ctx* ctx, aggmg_hier* h, const double* cur, double* alt, const double* b, int nsweeps, double alpha,
            CgtChk* chk = nullptr);
// compulsory bytes (every array of the launch once) of a chain level's fused launches / of a stand-alone sweep or residual launch
int cgt_launch_bytes(aggmg_ctx* ctx, const aggmg_hier* h, int level, bool down, bool up, bool has_x0, int64_t* rd, int64_t* wr);
int cgt_op_launch_bytes(const CgtDev& g, bool sweeps, int64_t* rd, int64_t* wr);

// ---------------------------------------------------------------------------------------------
// device-side set-up (setup.hip)
// ---------------------------------------------------------------------------------------------
int setup_csc_upload(aggmg_ctx* ctx, int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval,
                     const double* nzval, int one_based, CsrDev* out);
int op_ensure_csr(aggmg_ctx* ctx, aggmg_op* op);         // row-gather CSR + its CSR-stream row blocks
int setup_block_order(aggmg_ctx* ctx, aggmg_smoother* sm);   // blocks in ascending order of their smallest index, in place
int setup_block_cover(aggmg_ctx* ctx, aggmg_smoother* sm);   // row -> covering (block, local row) entries, on the device
int op_ensure_csc_blocks(aggmg_ctx* ctx, aggmg_op* op);  // CSR-stream row blocks of the transposed orientation
int op_host_csr(aggmg_ctx* ctx, aggmg_op* op, HostCsr* h);
int setup_jacobi_diag(aggmg_ctx* ctx, const aggmg_op* A, double** diag);
int setup_invert_blocks(aggmg_ctx* ctx, int64_t nb, int m, const double* blocks_dev, int colmajor, double* inv_dev,
                        int64_t* first_singular);
int setup_block_smoother(aggmg_ctx* ctx, aggmg_smoother* sm, const int64_t* blockinds, int one_based, int want_btd);
int setup_transfer_btd(aggmg_ctx* ctx, const aggmg_op* L, const BtdDev* Abtd, int mf, int64_t nef, int hint_mc,
                       TransferBtd* out, bool* ok);
int setup_cr(aggmg_ctx* ctx, const aggmg_op* Ac, int hint_m, CrDev* cr);
int cgt_detect(aggmg_ctx* ctx, aggmg_smoother* sm);   // chain form from the operator's own pattern (no element lists)
// chunk-interleaved boundary rows of the element-partitioned coarsest solve (aggmg_hip.hip; used by dist.hip)
int coarse_chunk_forward_interleaved(aggmg_ctx* ctx, aggmg_hier* h, const double* rhs_owned, int64_t blk_lo, int64_t blk_hi, double* Z);
int coarse_boundary_solve_interleaved(aggmg_ctx* ctx, aggmg_hier* h, const double* Z, double* xq);
void cr_discard(CrDev* cr);                                                  // frees the factors, valid = false
int setup_probe_vector(aggmg_ctx* ctx, int64_t n, double* w);                // hash-random entries in [-1, 1)
int setup_smooth_vector(aggmg_ctx* ctx, int64_t n, double* w);               // 1 + cos(pi i / n) / 2
int setup_band_matvec_add(aggmg_ctx* ctx, const aggmg_op* A, int m, const double* x, double sign, double* y);  // y += sign A x, A block-tridiagonal (block size m), deterministic
