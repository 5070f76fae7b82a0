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
template <int M>
struct CgtTile {
  static constexpr int NT = kThreads;
  static constexpr int NS = (M <= 4) ? 2 : 3;
  static constexpr int EPS = NT / M;
  static constexpr int TE = EPS * NS;
};

static int cgt_tile_blocks(int m) {
  switch (m) {
    case 1: return CgtTile<1>::TE;
    case 2: return CgtTile<2>::TE;
    case 3: return CgtTile<3>::TE;
    case 4: return CgtTile<4>::TE;
    case 5: return CgtTile<5>::TE;
    case 6: return CgtTile<6>::TE;
    case 7: return CgtTile<7>::TE;
    case 8: return CgtTile<8>::TE;
  }
  return 0;
}

// sweeps fused into one launch: every sweep costs one block of halo per side
static int cgt_max_sweeps(int m) { return std::max(1, std::min(8, cgt_tile_blocks(m) / 8)); }

template <int M>
static int cgt_launch_t(aggmg_ctx* ctx, CgtArgs a) {
  using T = CgtTile<M>;
  // halo: one block per sweep and side; the residual needs one more valid neighbour on both sides,
  // the restriction one more block of residual on the left (chain) or on the right (agglomerating)
  int hl = a.nsweeps, hr = a.nsweeps;
  if (a.do_residual) {
    hl += 1 + (a.tout.type == kTrChain ? 1 : 0);
    hr += 1 + (a.tout.type == kTrAgg ? 1 : 0);
  }
  const int align = a.tout.type == kTrAgg ? a.tout.rho : 1;
  const int owned = ((T::TE - hl - hr) / align) * align;
  if (owned <= 0) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "chain tile too small for the requested halo");
  a.owned = owned;
  a.halo_left = hl;
  a.tile_split = 0;
  a.tile_skip = 0;
  const int64_t ntiles = (a.lv.ne + owned - 1) / owned;
  if (ntiles == 0) return AGGMG_OK;
  if (ntiles >= ((int64_t)1 << 31)) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "grid too large");
  const size_t lds = (size_t)2 * (T::TE + 2) * M * sizeof(double);
  hipLaunchKernelGGL((cgt_fused_kernel<M, T::NS, T::NT>), dim3((unsigned)ntiles), dim3(T::NT), lds, ctx->stream, a);
  HIPCHK(hipGetLastError());
  return AGGMG_OK;
}

static int cgt_launch(aggmg_ctx* ctx, const CgtDev& g, const CgtArgs& a) {
  switch (g.m) {
#define CASE(MM) \
  case MM:       \
    return cgt_launch_t<MM>(ctx, a);
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  return fail(ctx, AGGMG_ERR_UNSUPPORTED, "chain block size not instantiated");
}

static CgtArgs cgt_args(const CgtDev& g) {
  CgtArgs a;
  std::memset(&a, 0, sizeof(a));
  a.lv = CgtLevel{g.dblk, g.subrow, g.supcol, g.ne};
  a.perm = g.perm;
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

// ---------------------------------------------------------------------------------------------
// set-up: element chain -> permutation -> packed operator
// ---------------------------------------------------------------------------------------------
int cgt_build(aggmg_ctx* ctx, aggmg_smoother* sm, const int64_t* elems, int64_t m1, int64_t nel, int one_based) {
  aggmg_op* A = sm->A;
  const int64_t N = A->m;
  const int64_t p = m1 - 1;
  if (p < 1 || p > 8 || nel < 1) return AGGMG_OK;  // generic path
  if (N != nel * p + 1) return AGGMG_OK;
  const int64_t base = one_based ? 1 : 0;
  const int m = (int)p;
  const int64_t ne = nel + 1, Np = ne * m;
  if (Np >= ((int64_t)1 << 31)) return AGGMG_OK;
  auto el = [&](int64_t e, int j) -> int64_t { return elems[e * m1 + j] - base; };
  auto g = std::make_shared<CgtDev>();
  g->m = m;
  g->ne = ne;
  g->N = N;
  g->h_perm.assign(Np, -1);
  g->h_inv.assign(N, -1);
  // block e = [first node of element e, its nodes 3..p+1]; node 2 must open the next element
  for (int64_t e = 0; e < nel; ++e) {
    for (int j = 0; j <= m; ++j) {
      const int64_t v = el(e, j);
      if (v < 0 || v >= N) return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_jacobi_setup_elements: node index out of range");
    }
    if (e + 1 < nel && el(e, 1) != el(e + 1, 0)) return AGGMG_OK;  // not a chain: generic path
    g->h_perm[e * m] = (int32_t)el(e, 0);
    for (int j = 1; j < m; ++j) g->h_perm[e * m + j] = (int32_t)el(e, j + 1);
  }
  g->h_perm[nel * m] = (int32_t)el(nel - 1, 1);
  for (int64_t q = 0; q < Np; ++q) {
    const int32_t o = g->h_perm[q];
    if (o < 0) continue;
    if (g->h_inv[o] >= 0) return AGGMG_OK;  // a node listed twice: not a chain
    g->h_inv[o] = (int32_t)q;
  }
  for (int64_t o = 0; o < N; ++o)
    if (g->h_inv[o] < 0) return AGGMG_OK;
  // pack: every stored entry must fall into the diagonal block, the sub-diagonal row or the
  // super-diagonal column of its block row
  const HostCsr& h = A->host;
  std::vector<double> dblk((size_t)Np * m, 0.0), subrow((size_t)Np, 0.0), supcol((size_t)Np, 0.0);
  std::atomic<int> bad{0};
  const std::vector<int32_t>& inv = g->h_inv;
  parallel_for(N, [&](int64_t rb, int64_t re) {
    for (int64_t r = rb; r < re && !bad.load(std::memory_order_relaxed); ++r) {
      const int64_t q = inv[r];
      const int64_t e = q / m;
      const int i = (int)(q - e * m);
      for (int32_t pp = h.rowptr[r]; pp < h.rowptr[r + 1]; ++pp) {
        const int64_t qc = inv[h.colind[pp]];
        const int64_t ce = qc / m;
        const int cj = (int)(qc - ce * m);
        const double v = h.vals[pp];
        if (ce == e) {
          dblk[q * m + cj] = v;
        } else if (ce == e - 1 && i == 0) {
          subrow[e * m + cj] = v;
        } else if (ce == e + 1 && cj == 0) {
          supcol[q] = v;
        } else if (v != 0.0) {
          bad.store(1);
          break;
        }
      }
    }
  });
  if (bad.load()) return AGGMG_OK;  // couplings beyond the chain pattern: generic path
  for (int64_t q = 0; q < Np; ++q)
    if (g->h_perm[q] < 0) dblk[q * m + (q % m)] = 1.0;  // padding rows: identity
  CHECK(dev_upload(ctx, dblk, &g->dblk));
  CHECK(dev_upload(ctx, subrow, &g->subrow));
  CHECK(dev_upload(ctx, supcol, &g->supcol));
  CHECK(dev_upload(ctx, g->h_perm, &g->perm));
  sm->cgt = g;
  A->cgt = g;
  return AGGMG_OK;
}

// ---------------------------------------------------------------------------------------------
// set-up: structured transfers of a chain level
// ---------------------------------------------------------------------------------------------
int cgt_build_transfer(aggmg_ctx* ctx, const aggmg_op* L, const CgtDev& f, const CgtDev* coarse, int hint_mc,
                       TransferCgt* out, bool* ok) {
  *ok = false;
  if (!L->host_valid) return AGGMG_OK;
  const HostCsr& h = L->host;
  const int64_t Nf = L->m, Nc = L->n;
  if (Nf != f.N) return AGGMG_OK;
  const int M = f.m;
  const int64_t nel = f.ne - 1, Np = f.ne * M;
  const int tile = cgt_tile_blocks(M);

  // ---- chain: the coarse level is a CG level on the same elements ------------------------------
  if (nel > 0 && Nc > 1 && (Nc - 1) % nel == 0 && (Nc - 1) / nel <= 8 && M >= 2) {
    const int mc = (int)((Nc - 1) / nel);
    std::vector<int32_t> cperm, cinv;
    bool have = false;
    if (coarse && coarse->N == Nc && coarse->m == mc && coarse->ne == f.ne) {
      cperm = coarse->h_perm;
      cinv = coarse->h_inv;
      have = true;
    } else if (!coarse) {
      // the coarse level carries no element lists (the coarsest level has no smoother): read its
      // chain off L -- a fine vertex row has ONE entry (its coarse vertex), the first interior row of
      // element e lists every coarse node of element e
      cperm.assign((size_t)f.ne * mc, -1);
      cinv.assign(Nc, -1);
      have = true;
      for (int64_t e = 0; e <= nel && have; ++e) {
        const int64_t r = f.h_perm[e * M];
        if (h.rowptr[r + 1] - h.rowptr[r] != 1) have = false;
        else cperm[e * mc] = h.colind[h.rowptr[r]];
      }
      for (int64_t e = 0; e < nel && have && mc > 1; ++e) {
        const int64_t r = f.h_perm[e * M + 1];
        int j = 1;
        for (int32_t pp = h.rowptr[r]; pp < h.rowptr[r + 1]; ++pp) {
          const int32_t c = h.colind[pp];
          if (c == cperm[e * mc] || c == cperm[(e + 1) * mc]) continue;
          if (j >= mc) {
            have = false;
            break;
          }
          cperm[e * mc + j++] = c;
        }
        if (j != mc) have = false;
      }
      for (int64_t q = 0; q < (int64_t)cperm.size() && have; ++q) {
        const int32_t o = cperm[q];
        if (o < 0) continue;
        if (o >= Nc || cinv[o] >= 0) have = false;
        else cinv[o] = (int32_t)q;
      }
      for (int64_t o = 0; o < Nc && have; ++o)
        if (cinv[o] < 0) have = false;
    }
    if (have) {
      const int w = mc + 1;
      std::vector<double> l((size_t)Np * w, 0.0);
      std::atomic<int> bad{0};
      parallel_for(Nf, [&](int64_t rb, int64_t re) {
        for (int64_t r = rb; r < re && !bad.load(std::memory_order_relaxed); ++r) {
          const int64_t q = f.h_inv[r];
          const int64_t e = q / M;
          for (int32_t pp = h.rowptr[r]; pp < h.rowptr[r + 1]; ++pp) {
            const int64_t qc = cinv[h.colind[pp]];
            const int64_t ce = qc / mc;
            const int cj = (int)(qc - ce * mc);
            if (ce == e)
              l[q * w + cj] = h.vals[pp];
            else if (ce == e + 1 && cj == 0)
              l[q * w + mc] = h.vals[pp];
            else if (h.vals[pp] != 0.0) {
              bad.store(1);
              break;
            }
          }
        }
      });
      if (!bad.load()) {
        CHECK(dev_upload(ctx, l, &out->l));
        CHECK(dev_upload(ctx, cperm, &out->cperm));
        out->type = kTrChain;
        out->mc = mc;
        out->rho = 1;
        out->nec = f.ne;
        *ok = true;
        return AGGMG_OK;
      }
    }
  }

  // ---- agglomerating: the coarse level has contiguous blocks of mc DoFs per rho fine elements ---
  for (int mc = 1; mc <= 16; ++mc) {
    if (hint_mc > 0 && mc != hint_mc) continue;
    if (Nc % mc) continue;
    const int64_t nec = Nc / mc;
    if (nec == 0 || nel % nec) continue;
    const int64_t rho = nel / nec;
    if (rho > 64 || rho * 4 > tile) continue;
    std::vector<double> l((size_t)Np * mc, 0.0), lp((size_t)f.ne * mc, 0.0);
    std::atomic<int> bad{0};
    parallel_for(Nf, [&](int64_t rb, int64_t re) {
      for (int64_t r = rb; r < re && !bad.load(std::memory_order_relaxed); ++r) {
        const int64_t q = f.h_inv[r];
        const int64_t e = q / M;
        const int i = (int)(q - e * M);
        const int64_t J = e / rho;
        for (int32_t pp = h.rowptr[r]; pp < h.rowptr[r + 1]; ++pp) {
          const int64_t c = h.colind[pp];
          const int64_t Jc = c / mc;
          const int cj = (int)(c - Jc * mc);
          if (Jc == J)
            l[q * mc + cj] = h.vals[pp];
          else if (i == 0 && Jc == J - 1 && e == J * rho)
            lp[e * mc + cj] = h.vals[pp];
          else if (h.vals[pp] != 0.0) {
            bad.store(1);
            break;
          }
        }
      }
    });
    if (bad.load()) continue;
    CHECK(dev_upload(ctx, l, &out->l));
    CHECK(dev_upload(ctx, lp, &out->lp));
    out->type = kTrAgg;
    out->mc = mc;
    out->rho = (int)rho;
    out->nec = nec;
    *ok = true;
    return AGGMG_OK;
  }
  return AGGMG_OK;
}

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
};

static int cgt_run(aggmg_ctx* ctx, const CgtDev& g, const CgtChain& c, double alpha, int nsweeps, const CgtArgs& first,
                   const CgtArgs& last, int kind, int level) {
  // at most smax sweeps per launch (2 * smax + 3 blocks of halo always fit a tile)
  const int smax = cgt_max_sweeps(g.m);
  const int nl = std::max(1, (nsweeps + smax - 1) / smax);
  if (nl > 1 && (!c.t0 || (nl > 2 && !c.t1))) return fail(ctx, AGGMG_ERR_UNSUPPORTED, "internal: chain temporaries missing");
  const double* src = c.src;
  bool src_ext = c.src_ext;
  int left = nsweeps;
  for (int q = 0; q < nl; ++q) {
    CgtArgs a = cgt_args(g);
    a.alpha = alpha;
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
    {
      ProfScope ps(ctx, q == nl - 1 ? kind : AGGMG_KIND_SMOOTH, level);
      CHECK(cgt_launch(ctx, g, a));
    }
    src = dst;
    src_ext = false;
  }
  return AGGMG_OK;
}

int cgt_smooth_ext(aggmg_ctx* ctx, const CgtDev& g, const double* u_in, const double* b, double alpha, int nsweeps,
                   double* u_out, int level) {
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
  if (nsweeps > cgt_max_sweeps(g.m)) {
    CHECK(scratch(ctx, 0, g.ne * g.m, &c.t0));
    CHECK(scratch(ctx, 1, g.ne * g.m, &c.t1));
  }
  CgtArgs none;
  std::memset(&none, 0, sizeof(none));
  return cgt_run(ctx, g, c, alpha, nsweeps, none, none, AGGMG_KIND_SMOOTH, level);
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
  CgtArgs none, last;
  std::memset(&none, 0, sizeof(none));
  std::memset(&last, 0, sizeof(last));
  last.do_residual = 1;
  last.tout = cgt_xfer(*l.tc, c.native_io);
  last.rc_out = c.rhs;
  return cgt_run(ctx, g, ch, alpha, nPre, none, last, AGGMG_KIND_FUSED_DOWN, k);
}

// ascending half (src/solvers.jl:41-47): prolongation-add, nPost sweeps
int cgt_up(aggmg_ctx* ctx, aggmg_hier* h, int k, const double* rhs, int nPost, double alpha, double* dst) {
  const int n = (int)h->lv.size();
  Level& l = h->lv[k];
  Level& c = h->lv[k + 1];
  const CgtDev& g = *l.S->cgt;
  CgtChain ch;
  ch.src = l.u[0];
  ch.src_ext = false;
  ch.b = rhs;
  ch.b_ext = (k == 0) || !l.native_io;
  ch.dst = dst;
  ch.dst_ext = (k == 0) || !l.native_io;
  // launch 0 consumes u[0] and writes tmp; launch 1 may then overwrite u[0], and so on
  ch.t0 = l.tmp;
  ch.t1 = l.u[0];
  CgtArgs first, none;
  std::memset(&first, 0, sizeof(first));
  std::memset(&none, 0, sizeof(none));
  first.tin = cgt_xfer(*l.tc, c.native_io);
  first.uc = (k + 1 == n - 1) ? c.u[0] : c.u[1];
  return cgt_run(ctx, g, ch, alpha, nPost, first, none, AGGMG_KIND_FUSED_UP, k);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int aggmg_jacobi_setup_elements(aggmg_ctx* ctx, aggmg_op* A, int64_t nodes_per_element, int64_t n_elements,
                                           const int64_t* element_nodes, int one_based, aggmg_smoother** out) {
  if (!ctx || !A || !out) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_jacobi_setup_elements: NULL argument");
  CHECK(aggmg_jacobi_setup(ctx, A, out));
  if (!element_nodes || n_elements <= 0 || nodes_per_element < 2) return AGGMG_OK;
  std::unique_ptr<aggmg_smoother> sm(*out);
  *out = nullptr;
  CHECK(cgt_build(ctx, sm.get(), element_nodes, nodes_per_element, n_elements, one_based));
  *out = sm.release();
  return AGGMG_OK;
}
