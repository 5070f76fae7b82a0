// CG chain path of libaggmg_hip: continuous-Galerkin levels renumbered element by element run a
// temporally blocked fused point-Jacobi kernel (cgt_kernels.hpp) instead of the generic CSR kernels.
//
// The reference smooths CG levels with JacobiSmoother (cg_smoother(cgMesh, A, :jac),
// src/smoother.jl:88-102; src/mesh_heirarchy.jl:51,58) on matrices numbered "vertices first"
// (src/cg_mesh.jl:37-45,59-65).  aggmg_jacobi_setup_elements receives the element node lists the
// reference's own cg_smoother has at hand (cgMesh.mElements[k].mNodesInd -- the mBlockInds matrix of
// its Schwarz variants, src/smoother.jl:104-134), derives the element-contiguous order from them and
// re-packs the operator -- every stored entry is read from the uploaded matrix, nothing is assumed
// about the mesh beyond "consecutive elements share exactly one node".
#include "internal.hpp"

// ---------------------------------------------------------------------------------------------
// tiles
// ---------------------------------------------------------------------------------------------
// slabs per thread (tile = NT / M * NS blocks): tuning knobs.  tools/exp_chain_tiles.sh on the config-5
// hierarchy at 2^24 elements: one slab everywhere 4.22 ms per V-cycle, two 4.31, four (M <= 2) 4.36-4.41 --
// occupancy beats the smaller halo share, as on the block-tridiagonal kernel
#ifndef AGGMG_CGT_NS1
#define AGGMG_CGT_NS1 1
#endif
#ifndef AGGMG_CGT_NS2
#define AGGMG_CGT_NS2 1
#endif
#ifndef AGGMG_CGT_NS4
#define AGGMG_CGT_NS4 1
#endif
// K: 0 point Jacobi, 1 element Schwarz, 2 red-black element Gauss-Seidel.  The Gauss-Seidel sweeps cost four blocks
// of halo each: two slabs per thread there (config-5 hierarchy at 2^24: 6.8 instead of 8.9 ms per V(3,3) cycle); the
// Schwarz sweeps (two blocks of halo) are faster with one (31 vs 34 us per sweep at 2^20 elements, p = 4)
#ifndef AGGMG_CGT_NS_GS
#define AGGMG_CGT_NS_GS 2
#endif
#ifndef AGGMG_CGT_NT
#define AGGMG_CGT_NT kThreads
#endif
template <int M, int K = 0>
struct CgtTile {
  static constexpr int NT = AGGMG_CGT_NT;   // threads per workgroup (tuning knob: tools/exp_tiles_r04.sh)
  static constexpr int NS0 = (M == 1) ? AGGMG_CGT_NS1 : (M == 2) ? AGGMG_CGT_NS2 : (M <= 4) ? AGGMG_CGT_NS4 : 3;
  static constexpr int NS = (K == 2) ? (NS0 < AGGMG_CGT_NS_GS ? AGGMG_CGT_NS_GS : NS0) : NS0;
  static constexpr int EPS = NT / M;
  static constexpr int TE = EPS * NS;
};

template <int SW>
static int cgt_tile_blocks_t(int m) {
  switch (m) {
    case 1: return CgtTile<1, SW>::TE;
    case 2: return CgtTile<2, SW>::TE;
    case 3: return CgtTile<3, SW>::TE;
    case 4: return CgtTile<4, SW>::TE;
    case 5: return CgtTile<5, SW>::TE;
    case 6: return CgtTile<6, SW>::TE;
    case 7: return CgtTile<7, SW>::TE;
    case 8: return CgtTile<8, SW>::TE;
  }
  return 0;
}
int cgt_tile_blocks(int m) { return cgt_tile_blocks_t<0>(m); }

// sweeps fused into one launch: every sweep costs one block of halo per side (element Schwarz sweeps: two,
// red-black element Gauss-Seidel: four)
static int cgt_halo_per_sweep(int sw) { return sw == 3 ? 4 : (sw ? 2 : 1); }
static int cgt_max_sweeps(int m, int sw = 0) {
  const int te = sw == 3 ? cgt_tile_blocks_t<2>(m) : cgt_tile_blocks_t<0>(m);
  return std::max(1, std::min(8, te / (8 * cgt_halo_per_sweep(sw))));
}

int cgt_max_fused_sweeps(const CgtDev& g) { return cgt_max_sweeps(g.m, g.sw); }

template <int M, int K>
static int cgt_launch_tt(aggmg_ctx* ctx, CgtArgs a, int sw, int64_t* ntiles_out);

template <int M>
static int cgt_launch_t(aggmg_ctx* ctx, CgtArgs a, int sw, int64_t* ntiles_out) {
  if (a.nsweeps == 0) sw = 0;
  if (sw == 3) return cgt_launch_tt<M, 2>(ctx, a, sw, ntiles_out);
  return sw ? cgt_launch_tt<M, 1>(ctx, a, sw, ntiles_out) : cgt_launch_tt<M, 0>(ctx, a, sw, ntiles_out);
}

template <int M, int K>
static int cgt_launch_tt(aggmg_ctx* ctx, CgtArgs a, int sw, int64_t* ntiles_out) {
  using T = CgtTile<M, K>;
  // halo: one block per sweep and side (element Schwarz: the update of a block reads the residual of its two
  // neighbours, i.e. the iterate two blocks away); the residual needs one more valid neighbour on both sides,
  // the restriction one more block of residual on the left (chain) or on the right (agglomerating)
  // (red-black element Gauss-Seidel: two such half-sweeps per sweep)
  int hl = a.nsweeps * cgt_halo_per_sweep(sw), hr = hl;
  if (sw == 3 && a.gs == 0) a.gs = 1;
  if (a.do_residual) {
    hl += 1 + (a.tout.type == kTrChain ? 1 : 0);
    hr += 1 + (a.tout.type == kTrAgg ? 1 : 0);
  } else if (a.chk_part) {   // a checkpoint after the last sweep forms residual rows of the final iterate
    hl += 1;
    hr += 1;
  }
  if (a.chk_part && K != 0) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "internal: checkpoint launch with element-block sweeps");
  const int align = a.tout.type == kTrAgg ? a.tout.rho : 1;
  const int owned = ((T::TE - hl - hr) / align) * align;
  if (owned <= 0) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "chain tile too small for the requested halo");
  a.owned = owned;
  a.halo_left = hl;
  a.tile_split = 0;
  a.tile_skip = 0;
  const int64_t ntiles = (a.lv.ne + owned - 1) / owned;
  if (ntiles_out) *ntiles_out = ntiles;
  if (ntiles == 0) return AGGMG_OK;
  if (ntiles >= ((int64_t)1 << 31)) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "grid too large");
  a.chk_tiles = ntiles;
  if (a.chk_stride < 1) a.chk_stride = 1 << 30;
  const size_t lds = (size_t)(sw ? 3 : 2) * (T::TE + 2) * M * sizeof(double) + (a.chk_part ? (size_t)2 * (T::NT / 64) * sizeof(double) : 0);
  if constexpr (K == 0) {
    if (a.chk_part) {
      hipLaunchKernelGGL((cgt_fused_kernel<M, T::NS, T::NT, 0, true>), dim3((unsigned)ntiles), dim3(T::NT), lds, ctx->stream, a);
      HIPCHK(hipGetLastError());
      return AGGMG_OK;
    }
  }
  if constexpr (K == 2) {
    hipLaunchKernelGGL((cgt_fused_kernel<M, T::NS, T::NT, 3>), dim3((unsigned)ntiles), dim3(T::NT), lds, ctx->stream, a);
  } else if constexpr (K == 1) {
    if (sw == 1)
      hipLaunchKernelGGL((cgt_fused_kernel<M, T::NS, T::NT, 1>), dim3((unsigned)ntiles), dim3(T::NT), lds, ctx->stream, a);
    else
      hipLaunchKernelGGL((cgt_fused_kernel<M, T::NS, T::NT, 2>), dim3((unsigned)ntiles), dim3(T::NT), lds, ctx->stream, a);
  } else {
    hipLaunchKernelGGL((cgt_fused_kernel<M, T::NS, T::NT, 0>), dim3((unsigned)ntiles), dim3(T::NT), lds, ctx->stream, a);
  }
  HIPCHK(hipGetLastError());
  return AGGMG_OK;
}

static int cgt_launch(aggmg_ctx* ctx, const CgtDev& g, const CgtArgs& a, int64_t* ntiles_out = nullptr) {
  switch (g.m) {
#define CASE(MM) \
  case MM:       \
    return cgt_launch_t<MM>(ctx, a, g.sw, ntiles_out);
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  return fail(ctx, AGGMG_ERR_UNSUPPORTED, "chain block size not instantiated");
}

static CgtArgs cgt_args(const CgtDev& g) {
  CgtArgs a;
  std::memset(&a, 0, sizeof(a));
  a.lv = CgtLevel{g.dblk, g.subrow, g.supcol, g.ne, g.zrows, g.zlast};
  a.perm = g.perm;
  a.affine = g.affine ? 1 : 0;
  return a;
}

static CgtXfer cgt_xfer(const TransferCgt& t, bool coarse_native) {
  CgtXfer x;
  x.type = t.type;
  x.l = t.l;
  x.lp = t.lp;
  x.cperm = (t.type == kTrChain && !coarse_native) ? t.cperm : nullptr;
  x.mc = t.mc;
  x.rho = t.rho;
  x.nec = t.nec;
  return x;
}

// (set-up of the chain form and of its transfers: setup.hip, on the device)

// ---------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------
// Chunk chain: src -(s1 sweeps)-> t0 -> t1 -> ... -> dst.  `first` decorates the first launch
// (prolongation), `last` the last one (residual / restriction); intermediates are block-ordered.
struct CgtChain {
  const double* src = nullptr;
  bool src_ext = false;
  const double* b = nullptr;
  bool b_ext = false;
  double* dst = nullptr;
  bool dst_ext = false;
  double *t0 = nullptr, *t1 = nullptr;  // block-ordered temporaries (needed when the sweeps do not fit one launch)
  int gs = 0;  // red-black element Gauss-Seidel: colour order of every sweep (1 forward, 2 reverse)
};

static int cgt_run(aggmg_ctx* ctx, const CgtDev& g, const CgtChain& c, double alpha, int nsweeps, const CgtArgs& first,
                   const CgtArgs& last, int kind, int level, CgtChk* chk = nullptr) {
  // at most smax sweeps per launch (2 * smax + 3 blocks of halo always fit a tile)
  const int smax = cgt_max_sweeps(g.m, g.sw);
  const int nl = std::max(1, (nsweeps + smax - 1) / smax);
  if (chk && (nl != 1 || g.sw != 0)) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "internal: checkpoints need the sweeps in one point-Jacobi launch");
  if (nl > 1 && (!c.t0 || (nl > 2 && !c.t1))) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "internal: chain temporaries missing");
  const double* src = c.src;
  bool src_ext = c.src_ext;
  int left = nsweeps;
  for (int q = 0; q < nl; ++q) {
    CgtArgs a = cgt_args(g);
    a.alpha = alpha;
    a.gs = c.gs;
    a.nsweeps = std::min(left, smax);
    left -= a.nsweeps;
    a.u_in = src;
    a.b = c.b;
    a.ext = (src && src_ext ? kExtUin : 0) | (c.b_ext ? kExtB : 0);
    if (q == 0) {
      a.tin = first.tin;
      a.uc = first.uc;
    }
    double* dst;
    if (q == nl - 1) {
      dst = c.dst;
      if (c.dst_ext) a.ext |= kExtUout;
      a.do_residual = last.do_residual;
      a.r_out = last.r_out;
      if (last.r_out && (last.ext & kExtRout)) a.ext |= kExtRout;
      a.tout = last.tout;
      a.rc_out = last.rc_out;
    } else {
      dst = (q % 2 == 0) ? c.t0 : c.t1;  // launch q reads what launch q - 1 wrote
    }
    a.u_out = dst;
    if (chk) {
      a.chk_sweep = chk->sweep;
      a.chk_stride = chk->stride;
      a.chk_final = chk->final;
      a.chk_exact = chk->exact;
      a.chk_part = chk->part;
      a.ext |= kExtB;   // (the caller-side indices are formed: chk_exact goes through them)
      if (!c.b_ext) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "internal: checkpoint launch on block-ordered vectors");
    }
    {
      ProfScope ps(ctx, q == nl - 1 ? kind : AGGMG_KIND_SMOOTH, level);
      CHECK(cgt_launch(ctx, g, a, chk ? &chk->ntiles : nullptr));
    }
    src = dst;
    src_ext = false;
  }
  return AGGMG_OK;
}

int cgt_smooth_ext(aggmg_ctx* ctx, const CgtDev& g, const double* u_in, const double* b, double alpha, int nsweeps,
                   double* u_out, int level, CgtChk* chk) {
  if (nsweeps == 0) {
    if (u_in)
      HIPCHK(hipMemcpyAsync(u_out, u_in, g.N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    else
      HIPCHK(hipMemsetAsync(u_out, 0, g.N * sizeof(double), ctx->stream));
    return AGGMG_OK;
  }
  CgtChain c;
  c.src = u_in;
  c.src_ext = true;
  c.b = b;
  c.b_ext = true;
  c.dst = u_out;
  c.dst_ext = true;
  if (nsweeps > cgt_max_sweeps(g.m, g.sw)) {
    CHECK(scratch(ctx, 0, g.ne * g.m, &c.t0));
    CHECK(scratch(ctx, 1, g.ne * g.m, &c.t1));
  }
  CgtArgs none;
  std::memset(&none, 0, sizeof(none));
  return cgt_run(ctx, g, c, alpha, nsweeps, none, none, AGGMG_KIND_SMOOTH, level, chk);
}

int cgt_residual_ext(aggmg_ctx* ctx, const CgtDev& g, const double* u, const double* b, double* r_out) {
  CgtArgs a = cgt_args(g);
  a.u_in = u;
  a.b = b;
  a.ext = kExtUin | kExtB | kExtRout;
  a.do_residual = 1;
  a.r_out = r_out;
  return cgt_launch(ctx, g, a);
}

// descending half on a fused chain level (src/solvers.jl:28-37): nPre sweeps, residual, restriction
int cgt_down(aggmg_ctx* ctx, aggmg_hier* h, int k, const double* uin, const double* rhs, int nPre, double alpha) {
  Level& l = h->lv[k];
  Level& c = h->lv[k + 1];
  const CgtDev& g = *l.S->cgt;
  CgtChain ch;
  ch.src = uin;
  ch.src_ext = true;  // only level 0 has an initial guess, and it is the caller's vector
  ch.b = rhs;
  ch.b_ext = (k == 0) || !l.native_io;
  ch.dst = l.u[0];
  ch.t0 = l.u[1];
  ch.t1 = l.tmp;
  ch.gs = 1;  // pre-smoothing: even elements, then odd ones
  CgtArgs none, last;
  std::memset(&none, 0, sizeof(none));
  std::memset(&last, 0, sizeof(last));
  last.do_residual = 1;
  last.tout = cgt_xfer(*l.tc, c.native_io);
  last.rc_out = c.rhs;
  return cgt_run(ctx, g, ch, alpha, nPre, none, last, AGGMG_KIND_FUSED_DOWN, k);
}

// ascending half (src/solvers.jl:41-47): prolongation-add, nPost sweeps
// src: the pre-smoothed iterate in block order (default: the level's u[0])
int cgt_up(aggmg_ctx* ctx, aggmg_hier* h, int k, const double* rhs, int nPost, double alpha, double* dst, const double* src,
           CgtChk* chk) {
  const int n = (int)h->lv.size();
  Level& l = h->lv[k];
  Level& c = h->lv[k + 1];
  const CgtDev& g = *l.S->cgt;
  CgtChain ch;
  ch.src = src ? src : l.u[0];
  ch.src_ext = false;
  ch.b = rhs;
  ch.b_ext = (k == 0) || !l.native_io;
  ch.dst = dst;
  ch.dst_ext = (k == 0) || !l.native_io;
  // launch 0 consumes the source and writes tmp; launch 1 may then overwrite the source, and so on
  ch.t0 = l.tmp;
  ch.t1 = const_cast<double*>(ch.src);
  ch.gs = 2;  // post-smoothing in the reverse colour order: the cycle stays symmetric
  CgtArgs first, none;
  std::memset(&first, 0, sizeof(first));
  std::memset(&none, 0, sizeof(none));
  first.tin = cgt_xfer(*l.tc, c.native_io);
  first.uc = (k + 1 == n - 1) ? c.u[0] : c.u[1];
  return cgt_run(ctx, g, ch, alpha, nPost, first, none, AGGMG_KIND_FUSED_UP, k, chk);
}

// Between two cycles of multigrid()'s loop (src/solvers.jl:124-126) the fine level post-smooths and then
// pre-smooths the same iterate with the same right-hand side: prolongation-add, nPost + nPre sweeps,
// residual and restriction in ONE launch -- the fine operator is read once per cycle instead of twice.
// cur -> alt, both in block order; b is the caller's vector.
int cgt_mid(aggmg_ctx* ctx, aggmg_hier* h, const double* cur, double* alt, const double* b, int nsweeps, double alpha,
            CgtChk* chk) {
  const int n = (int)h->lv.size();
  Level& l = h->lv[0];
  Level& c = h->lv[1];
  const CgtDev& g = *l.S->cgt;
  CgtChain ch;
  ch.src = cur;
  ch.src_ext = false;
  ch.b = b;
  ch.b_ext = true;
  ch.dst = alt;
  ch.dst_ext = false;
  ch.t0 = l.tmp;
  if (nsweeps > 2 * cgt_max_sweeps(g.m, g.sw)) CHECK(scratch(ctx, 0, g.ne * g.m, &ch.t1));
  CgtArgs first, last;
  std::memset(&first, 0, sizeof(first));
  std::memset(&last, 0, sizeof(last));
  first.tin = cgt_xfer(*l.tc, c.native_io);
  first.uc = (1 == n - 1) ? c.u[0] : c.u[1];
  last.do_residual = 1;
  last.tout = cgt_xfer(*l.tc, c.native_io);
  last.rc_out = c.rhs;
  return cgt_run(ctx, g, ch, alpha, nsweeps, first, last, AGGMG_KIND_FUSED_MID, 0, chk);
}

// ---- compulsory bytes: what the arrays of a launch hold, each read or written once -------------
static int64_t cgt_operator_bytes(const CgtDev& g, bool sweeps) {
  const int64_t D = sizeof(double), rows = g.ne * g.m;
  int64_t r = rows * g.m * D + g.ne * g.m * D + rows * D;    // dblk, subrow, supcol
  if (sweeps && g.sw) r += rows * (g.m + 1) * D + g.ne * (g.m + 1) * D;   // element inverses (Schwarz / element Gauss-Seidel)
  return r;
}
static int64_t cgt_xfer_bytes(const TransferCgt& t, const CgtDev& g) {
  const int64_t D = sizeof(double), rows = g.ne * g.m;
  return t.type == kTrChain ? rows * (t.mc + 1) * D : rows * t.mc * D + g.ne * t.mc * D;
}

int cgt_op_launch_bytes(const CgtDev& g, bool sweeps, int64_t* rd, int64_t* wr) {
  const int64_t D = sizeof(double);
  *rd = cgt_operator_bytes(g, sweeps) + 2 * g.N * D + (g.affine ? 0 : g.ne * g.m * 4);   // u and b in the caller's numbering
  *wr = g.N * D;
  return AGGMG_OK;
}

int cgt_launch_bytes(aggmg_ctx* ctx, const aggmg_hier* h, int level, bool down, bool up, bool has_x0, int64_t* rd, int64_t* wr) {
  const Level& l = h->lv[level];
  const Level& c = h->lv[level + 1];
  if (!l.S || !l.S->cgt || !l.tc) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "aggmg_hier_launch_bytes: not a fused chain level");
  const CgtDev& g = *l.S->cgt;
  const int64_t D = sizeof(double), rows = g.ne * g.m;
  const bool ext = (level == 0) || !l.native_io;
  const int64_t vec = ext ? g.N : rows;
  int64_t r = cgt_operator_bytes(g, true) + vec * D, w = 0;            // operator, right-hand side
  if (up || has_x0) r += (up ? rows : g.N) * D;                        // iterate in (block order inside the hierarchy)
  if (ext && !g.affine) r += rows * 4;                                 // block order -> caller's numbering
  r += cgt_xfer_bytes(*l.tc, g);                                       // rows of L (either direction; once when both)
  const int64_t nc = l.tc->nec * l.tc->mc;
  const bool cp = l.tc->type == kTrChain && !c.native_io && l.tc->cperm;
  if (up) r += nc * D + (cp ? nc * 4 : 0);                             // coarse iterate
  if (down) w += nc * D;                                               // restricted residual
  if (down && !up && cp) r += nc * 4;
  w += (up && !down ? vec : rows) * D;                                 // iterate out
  *rd = r;
  *wr = w;
  return AGGMG_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int aggmg_jacobi_setup_elements(aggmg_ctx* ctx, aggmg_op* A, int64_t nodes_per_element, int64_t n_elements,
                                           const int64_t* element_nodes, int one_based, aggmg_smoother** out) {
  if (!ctx || !A || !out) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_jacobi_setup_elements: NULL argument");
  const bool lists = element_nodes && n_elements > 0 && nodes_per_element >= 2;
  const bool detect = ctx->detect_chain;
  if (lists) ctx->detect_chain = false;   // the caller's lists decide, not the pattern
  const int st0 = aggmg_jacobi_setup(ctx, A, out);
  ctx->detect_chain = detect;
  CHECK(st0);
  if (!lists) return AGGMG_OK;
  std::unique_ptr<aggmg_smoother> sm(*out);
  *out = nullptr;
  CHECK(cgt_build(ctx, sm.get(), element_nodes, nodes_per_element, n_elements, one_based));
  *out = sm.release();
  return AGGMG_OK;
}
