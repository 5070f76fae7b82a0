// Fused point-Jacobi kernel for continuous-Galerkin (CG) levels in element-contiguous order.
//
// The reference numbers a CgMesh "all vertices first, then the p-1 interior nodes of every element"
// (src/cg_mesh.jl:37-45,59-65).  Renumbered element by element -- block e = [left vertex of
// element e, its interior nodes], plus one trailing block holding the last vertex -- every CG
// stiffness matrix (assembled or Galerkin-coarsened, src/mesh_heirarchy.jl:53-59) is
// block-tridiagonal with m = p rows per block, and its off-diagonal blocks are thin:
//
//     Sub_e = A[block e, block e-1]  has ONE non-zero ROW    (row 0: vertex e sees all of element e-1)
//     Sup_e = A[block e, block e+1]  has ONE non-zero COLUMN (column 0: element e sees vertex e+1)
//
// (the mirror image of the nodal-DG pattern of kernels.hpp).  Stored per level, fp64, no indices:
//     dblk   [ne*M][M]  rows of the diagonal blocks        (the Jacobi diagonal is dblk[r][r % M])
//     subrow [ne][M]    A[(e,0), (e-1, :)]
//     supcol [ne*M]     A[(e,i), (e+1, 0)]
// One workgroup owns a tile of blocks plus a halo, keeps the iterate in LDS (ping-pong) and its
// operator rows in registers and runs
//     [u += L uc]  ->  S sweeps  u += alpha * (b - A u) / diag(A)   ->  [r = b - A u  ->  rc = L' r]
// on ONE pass over HBM (JacobiSmoother src/smoother.jl:52-58 inside the loop of src/solvers.jl:32-46).
// The caller's vectors stay in the reference numbering: rows are fetched / stored through the
// level's permutation where a vector crosses the library boundary, so no permutation pass exists.
#pragma once
#include <type_traits>

#include "kernels.hpp"

namespace aggmg {

struct CgtLevel {
  const double *dblk, *subrow, *supcol;
  int64_t ne;  // blocks, the trailing (partial, identity-padded) one included
  // element Schwarz sweeps (SW != 0): rows of the element inverses, chain-local order (CgtDev::zrows / zlast)
  const double *zrows, *zlast;
};

// which vectors of a launch live in the caller's numbering (indexed through perm)
enum CgtExt : int { kExtUin = 1, kExtB = 2, kExtUout = 4, kExtRout = 8 };

// structured transfers of a CG level (rows of L grouped per fine block; fine row r = e*M + i)
//  chain  (CG p_hi -> CG p_lo, cg_cg_interpolation src/interpolation.jl:5-55): fine block e sees the
//         mc DoFs of coarse block e and the first DoF of coarse block e+1 (the shared vertex):
//         l [N][mc + 1]
//  agg    (CG -> DG / agglomerated DG, dg_cg_ / aggdg_cg_interpolation :145-220,:330-410): fine block e
//         sees coarse block J = e / rho; its vertex row also sees block J - 1 when e starts an
//         agglomerate:  l [N][mc],  lp [ne][mc]
enum CgtTransfer : int { kTrNone = 0, kTrChain = 1, kTrAgg = 2 };
constexpr int kCgtMaxMcUnrolled = 4;   // coarse block sizes whose prolongation loads are issued in one batch

struct CgtXfer {
  int type;
  const double* l;
  const double* lp;
  const int32_t* cperm;  // chain only: coarse vector in the caller's numbering (null: block order)
  int mc, rho;
  int64_t nec;  // coarse blocks
};

struct CgtArgs {
  CgtLevel lv;
  const int32_t* perm;  // [ne*M] block order -> caller's numbering, -1 for padding rows
  int affine;           // the order is the reference's own (vertices 0..ne-1, then M-1 interior nodes per element,
                        // src/cg_mesh.jl:37-45,59-65): caller-side indices are computed instead of read from perm
  int ext;              // CgtExt mask
  const double* u_in;   // nullptr: iterate starts at zero (src/solvers.jl:29-31)
  const double* b;
  double* u_out;        // nullptr: iterate not stored
  double alpha;
  int nsweeps;
  CgtXfer tin;          // prolongation-add before the sweeps (src/solvers.jl:42)
  const double* uc;
  int do_residual;      // residual after the sweeps (src/solvers.jl:36)
  double* r_out;
  CgtXfer tout;         // restriction of that residual
  double* rc_out;
  int owned, halo_left;
  int tile_split;
  int64_t tile_skip;
  int gs;  // SW == 3: colour order of a sweep, 1 = even elements then odd ones, 2 = the reverse (post-smoothing)
  // checkpoints (CHK variant; the fields and their meaning as FusedArgs's, kernels.hpp): after chk_sweep, chk_sweep +
  // chk_stride, ... sweeps, and (chk_final) after the last, the tile's sums of squares of b - A u and of u - chk_exact over
  // its owned rows go to chk_part[(k * chk_tiles + tile)][2]; chk_exact lives in the caller's numbering (through perm,
  // like b)
  int chk_sweep, chk_stride, chk_final;
  int64_t chk_tiles;
  const double* chk_exact;
  double* chk_part;
};

__device__ __forceinline__ int64_t cgt_tile(const CgtArgs& a) {
  return (int64_t)blockIdx.x + ((int)blockIdx.x >= a.tile_split ? a.tile_skip : 0);
}

// SW = 0: point-Jacobi sweeps.  SW = 1 / 2: additive / hybrid Schwarz sweeps over the ELEMENTS of the chain
// (AdditiveSchwarzSmoother, HybridSchwarzSmoother, src/smoother.jl:1-46; cg_smoother :104-134):
//     u += alpha * [1/count] * sum_e R_e' (A[nodes_e, nodes_e] \ R_e (b - A u))
// element e = block e plus the first row of block e + 1 (its right vertex); a vertex receives the
// contribution of the element on its left first, then its own element's (the reference's loop order);
// count = 2 on interior vertices, 1 elsewhere.  Two LDS phases per sweep (residual, then the element
// solves on it) and two blocks of halo per sweep and side.
// SW = 3 (EXTENSION, no reference counterpart -- SURVEY D1; BASELINE.json names a block-GS smoother for the
// CG-fine hierarchy of config 5): red-black element Gauss-Seidel.  A sweep is two half-sweeps, one per element
// colour (elements of one colour share no node): r = b - A u, then u[nodes_e] += alpha (A_e \ r[nodes_e]) for
// every element of the colour; a.gs gives the colour order.  Four blocks of halo per sweep and side.
template <int M, int NS, int NT, int SW = 0, bool CHK = false>
__global__ __launch_bounds__(NT) void cgt_fused_kernel(CgtArgs a) {
  static_assert(!CHK || SW == 0, "the checkpoint variant is for point-Jacobi launches");
  // GRP: a block's rows sit in M = 2^k adjacent lanes; lane i keeps entry i of the block's
  // sub-diagonal row and the dot product with the left neighbour is a cross-lane sum
  constexpr bool GRP = (M == 1 || M == 2 || M == 4 || M == 8);
  constexpr int EPS = NT / M;
  constexpr int TE = EPS * NS;
  extern __shared__ double lds[];
  double* buf0 = lds + M;  // index x*M + j, x in [-1, TE]
  double* buf1 = lds + (TE + 2) * M + M;
  double* rbuf = lds + 2 * (TE + 2) * M + M;  // SW: the residual of the current sweep

  const int tid = threadIdx.x;
  const bool active = tid < EPS * M;
  const int le = tid / M;
  const int i = tid - le * M;
  const int64_t ne = a.lv.ne;
  const int64_t e0 = cgt_tile(a) * a.owned - a.halo_left;

  if (tid < M) {
    buf0[-M + tid] = 0.0;
    buf0[TE * M + tid] = 0.0;
    buf1[-M + tid] = 0.0;
    buf1[TE * M + tid] = 0.0;
    if (SW) {
      rbuf[-M + tid] = 0.0;
      rbuf[TE * M + tid] = 0.0;
    }
  }

  double zr[NS][SW ? M + 1 : 1], zl[NS][SW ? M + 1 : 1];  // SW: own row of the element inverse; lane 0: the left element's last row
  double d[NS][M], sup[NS], sr[NS][GRP ? 1 : M], dg[NS], bb[NS], uu[NS];
  int32_t pr[NS];
  bool valid[NS];

  // ---- load phase ------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int x = s * EPS + le;
    const int64_t e = e0 + x;
    valid[s] = active && e >= 0 && e < ne;
    const int64_t row = e * M + i;
    uu[s] = 0.0;
    bb[s] = 0.0;
    sup[s] = 0.0;
    dg[s] = 1.0;
    pr[s] = -1;
#pragma unroll
    for (int j = 0; j < M; ++j) d[s][j] = 0.0;
#pragma unroll
    for (int j = 0; j < (GRP ? 1 : M); ++j) sr[s][j] = 0.0;
#pragma unroll
    for (int j = 0; j < (SW ? M + 1 : 1); ++j) {
      zr[s][j] = 0.0;
      zl[s][j] = 0.0;
    }
    if (valid[s]) {
      if (SW && a.nsweeps > 0) {
#pragma unroll
        for (int j = 0; j <= M; ++j) zr[s][j] = AGGMG_LD(a.lv.zrows[row * (M + 1) + j]);
        if (i == 0) {
#pragma unroll
          for (int j = 0; j <= M; ++j) zl[s][j] = a.lv.zlast[e * (M + 1) + j];
        }
      }
#pragma unroll
      for (int j = 0; j < M; ++j) d[s][j] = AGGMG_LD(a.lv.dblk[row * M + j]);
#pragma unroll
      for (int j = 0; j < M; ++j)
        if (j == i) dg[s] = d[s][j];
      sup[s] = AGGMG_LD(a.lv.supcol[row]);
      if (GRP) {
        sr[s][0] = AGGMG_LD(a.lv.subrow[row]);  // entry i of the block's sub-diagonal row
      } else if (i == 0) {
#pragma unroll
        for (int j = 0; j < (GRP ? 1 : M); ++j) sr[s][j] = a.lv.subrow[e * M + j];
      }
      if (a.ext) {
        if (a.affine)
          pr[s] = i == 0 ? (int32_t)e : (e == ne - 1 ? -1 : (int32_t)(ne + e * (M - 1) + (i - 1)));
        else
          pr[s] = a.perm[row];
      }
      const int64_t rb = (a.ext & kExtB) ? (int64_t)pr[s] : row;
      if (rb >= 0) bb[s] = a.b[rb];
      if (a.u_in) {
        const int64_t ru = (a.ext & kExtUin) ? (int64_t)pr[s] : row;
        if (ru >= 0) uu[s] = a.u_in[ru];
      }
      if (a.tin.type == kTrChain) {
        const int mc = a.tin.mc;
        const double* lr = a.tin.l + row * (mc + 1);
        double add = 0.0;
        // every load of the row's mc + 1 products issued before the first is used (a loop with a run-time trip count
        // takes them one dependent round trip at a time: the ascent was 8 % slower than the descent on the same bytes)
        auto batch = [&](auto cmax_tag) {
          constexpr int CMAX = decltype(cmax_tag)::value;
          double lv[CMAX + 1], uv[CMAX + 1];
#pragma unroll
          for (int c = 0; c <= CMAX; ++c) {
            lv[c] = 0.0;
            uv[c] = 0.0;
            if (c <= mc) {
              const int64_t cb = c < mc ? e * mc + c : (e + 1) * mc;   // coarse block e, DoF c; the last entry is DoF 0 of block e + 1
              if (cb < a.tin.nec * mc) {
                const int64_t ci = a.tin.cperm ? (int64_t)a.tin.cperm[cb] : cb;
                if (ci >= 0) {
                  lv[c] = lr[c];
                  uv[c] = a.uc[ci];
                }
              }
            }
          }
#pragma unroll
          for (int c = 0; c <= CMAX; ++c)
            if (c <= mc) add += lv[c] * uv[c];   // (a skipped entry adds 0.0 * 0.0: the sum is unchanged)
        };
        // (measured on the config-5 hierarchy at 2^24: p = 4 level ascent 1.171 -> 1.15 ms; the p = 2 level, two entries per
        // row, is 2 % slower with it: blocks of 4 rows and more only)
        if (M >= 4 && mc <= 2) {
          batch(std::integral_constant<int, 2>());
        } else if (M >= 4 && mc <= kCgtMaxMcUnrolled) {
          batch(std::integral_constant<int, kCgtMaxMcUnrolled>());
        } else {
          for (int c = 0; c <= mc; ++c) {
            const int64_t cb = c < mc ? e * mc + c : (e + 1) * mc;
            if (cb >= a.tin.nec * mc) continue;
            const int64_t ci = a.tin.cperm ? (int64_t)a.tin.cperm[cb] : cb;
            if (ci >= 0) add += lr[c] * a.uc[ci];
          }
        }
        uu[s] += add;
      } else if (a.tin.type == kTrAgg) {
        const int mc = a.tin.mc;
        const int64_t J = e / a.tin.rho;
        double add = 0.0;
        if (i == 0 && J >= 1 && e == J * a.tin.rho)
          for (int c = 0; c < mc; ++c) add += a.tin.lp[e * mc + c] * a.uc[(J - 1) * mc + c];
        if (J < a.tin.nec)
          for (int c = 0; c < mc; ++c) add += a.tin.l[row * mc + c] * a.uc[J * mc + c];
        uu[s] += add;
      }
    }
    if (active) buf0[x * M + i] = uu[s];
  }
  __syncthreads();

  // A u for this thread's row out of the LDS iterate `cur`, in ascending column order of the
  // reference numbering (left block's vertex, own vertex, right vertex, then interior nodes)
  auto row_Au = [&](int s, int x, const double* cur) -> double {
    const double* um = cur + (x - 1) * M;
    const double* ux = cur + x * M;
    const double* up = cur + (x + 1) * M;
    double t;
    if (GRP) {
      t = group_sum<M>(sr[s][0] * um[i]);
      if (i != 0) t = 0.0;
    } else {
      t = 0.0;
      if (i == 0) {
#pragma unroll
        for (int j = 0; j < (GRP ? 1 : M); ++j) t += sr[s][j] * um[j];
      }
    }
    t += d[s][0] * ux[0];
    t += sup[s] * up[0];
#pragma unroll
    for (int j = 1; j < M; ++j) t += d[s][j] * ux[j];
    return t;
  };

  // ---- sweeps: u <- u + alpha * ((b - A u) / diag)   (LDS ping-pong) ------------------------------
  double* cur = buf0;
  double* nxt = buf1;
  // CHK: residual rows of the iterate in `it` on the owned blocks (the operator rows are in registers: the expressions of
  // the closing residual), sums over the tile in a fixed order -- waves, then the workgroup's waves
  [[maybe_unused]] int kchk = 0;
  [[maybe_unused]] auto checkpoint = [&](const double* it) {
    double sr2 = 0.0, se2 = 0.0;
    const int c0 = a.halo_left, c1 = a.halo_left + a.owned;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      if (active) {
        const double t = row_Au(s, x, it);   // (every lane of a group takes part in the cross-lane sum)
        if (valid[s] && x >= c0 && x < c1) {
          const double r = bb[s] - t;
          sr2 += r * r;
          if (a.chk_exact && pr[s] >= 0) {   // (padding rows of the trailing block have no caller-side entry)
            const double dd = uu[s] - a.chk_exact[pr[s]];
            se2 += dd * dd;
          }
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sr2 += __shfl_xor(sr2, off, 64);
      se2 += __shfl_xor(se2, off, 64);
    }
    double* red = lds + 2 * (TE + 2) * M;   // (the launch reserves 2 * NT / 64 doubles behind the iterate buffers)
    if ((tid & 63) == 0) {
      red[2 * (tid >> 6)] = sr2;
      red[2 * (tid >> 6) + 1] = se2;
    }
    __syncthreads();
    if (tid == 0) {
      double tr = 0.0, te = 0.0;
      for (int w = 0; w < NT / 64; ++w) {
        tr += red[2 * w];
        te += red[2 * w + 1];
      }
      double* out = a.chk_part + (kchk * a.chk_tiles + cgt_tile(a)) * 2;
      out[0] = tr;
      out[1] = te;
    }
    ++kchk;
  };
  [[maybe_unused]] auto chk_due = [&](int sw) { return sw >= a.chk_sweep && (sw - a.chk_sweep) % a.chk_stride == 0; };
  for (int sw = 0; sw < (SW == 3 ? 2 * a.nsweeps : a.nsweeps); ++sw) {
    if constexpr (CHK) {
      if (chk_due(sw)) checkpoint(cur);
    }
    if (SW) {
      const int colour = (SW == 3) ? ((a.gs == 2) ? 1 - (sw & 1) : (sw & 1)) : 0;
      // phase A: the residual of every row of the tile into LDS
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int x = s * EPS + le;
        if (active) {
          const double r = bb[s] - row_Au(s, x, cur);
          rbuf[x * M + i] = valid[s] ? r : 0.0;
        }
      }
      __syncthreads();
      // phase B: row i of  A_e \ r_e  (own element), lane 0 also the last row of the left element's solve
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int x = s * EPS + le;
        if (active) {
          const double* rx = rbuf + x * M;
          const int64_t e = e0 + x;
          double y = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) y += zr[s][j] * rx[j];
          y += zr[s][M] * rx[M];  // the element's right vertex = first row of the next block
          if (SW == 3 && ((e & 1) != colour)) y = 0.0;  // not this element's half-sweep
          if (i == 0) {
            double yl = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) yl += zl[s][j] * rx[j - M];
            yl += zl[s][M] * rx[0];
            if (SW == 3 && (((e - 1) & 1) != colour)) yl = 0.0;
            y = yl + y;             // element e - 1 reaches the vertex before element e does
            if (SW == 2) {
              if (e > 0 && e < ne - 1) y = y / 2.0;  // mCountingMatrix: two elements share an interior vertex
            }
          }
          double un = uu[s] + a.alpha * y;
          if (!valid[s]) un = 0.0;
          uu[s] = un;
          nxt[x * M + i] = un;
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int x = s * EPS + le;
        if (active) {
          const double r = bb[s] - row_Au(s, x, cur);
          const double y = r / dg[s];
          double un = uu[s] + a.alpha * y;
          if (!valid[s]) un = 0.0;
          uu[s] = un;
          nxt[x * M + i] = un;
        }
      }
    }
    __syncthreads();
    double* t = cur;
    cur = nxt;
    nxt = t;
  }

  if constexpr (CHK) {
    if (a.chk_final || chk_due(a.nsweeps)) checkpoint(cur);   // a launch that ends on a checked iterate
  }

  // ---- store the iterate of the owned blocks -------------------------------------------------------
  const int xo0 = a.halo_left, xo1 = a.halo_left + a.owned;
  if (a.u_out) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      if (valid[s] && x >= xo0 && x < xo1) {
        const int64_t ro = (a.ext & kExtUout) ? (int64_t)pr[s] : (e0 + x) * M + i;
        if (ro >= 0) AGGMG_ST(a.u_out[ro], uu[s]);
      }
    }
  }
  if (!a.do_residual) return;

  // ---- residual r = b - A u: owned blocks, plus the one block whose rows the restriction of an
  // owned coarse block also touches (chain: the block on the left, agg: the vertex on the right) ----
  const int xr0 = xo0 - (a.tout.type == kTrChain ? 1 : 0);
  const int xr1 = xo1 + (a.tout.type == kTrAgg ? 1 : 0);
  double rr[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int x = s * EPS + le;
    rr[s] = 0.0;
    if (active) {
      const double t = row_Au(s, x, cur);  // every lane of a group takes part in the cross-lane sum
      if (valid[s] && x >= xr0 && x < xr1) {
        rr[s] = bb[s] - t;
        if (a.r_out && x >= xo0 && x < xo1) {
          const int64_t ro = (a.ext & kExtRout) ? (int64_t)pr[s] : (e0 + x) * M + i;
          if (ro >= 0) a.r_out[ro] = rr[s];
        }
      }
    }
  }
  if (a.tout.type == kTrNone) return;

  // ---- restriction rc = L' r: r through LDS, one thread per owned coarse DoF ------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int x = s * EPS + le;
    if (active) nxt[x * M + i] = rr[s];
  }
  __syncthreads();
  const int mc = a.tout.mc;
  if (a.tout.type == kTrChain) {
    const double* L = a.tout.l;
    const int w = mc + 1;
    for (int t = tid; t < a.owned * mc; t += NT) {
      const int xl = t / mc, c = t - xl * mc;
      const int x = xo0 + xl;
      const int64_t J = e0 + x;  // coarse block = fine block
      if (J >= ne || J >= a.tout.nec) continue;
      const int64_t rowb = J * M;
      // ascending fine row of the reference numbering: own vertex, the left element's rows, own interior
      double acc = L[rowb * w + c] * nxt[x * M];
      if (c == 0 && J > 0)
        for (int k = 0; k < M; ++k) acc += L[(rowb - M + k) * w + mc] * nxt[(x - 1) * M + k];
      for (int k = 1; k < M; ++k) acc += L[(rowb + k) * w + c] * nxt[x * M + k];
      const int64_t cb = J * mc + c;
      const int64_t ci = a.tout.cperm ? (int64_t)a.tout.cperm[cb] : cb;
      if (ci >= 0) a.rc_out[ci] = acc;
    }
  } else {
    const int rho = a.tout.rho;
    const int ncoarse = a.owned / rho;
    const int64_t J0 = (cgt_tile(a) * a.owned) / rho;
    for (int t = tid; t < ncoarse * mc; t += NT) {
      const int Jl = t / mc, c = t - Jl * mc;
      const int64_t J = J0 + Jl;
      if (J >= a.tout.nec) continue;
      const int xb = xo0 + Jl * rho;
      const int64_t rowb = (e0 + xb) * (int64_t)M;
      double acc = 0.0;
      for (int k = 0; k < rho * M; ++k) acc += a.tout.l[(rowb + k) * mc + c] * nxt[xb * M + k];
      // the vertex that closes the agglomerate on the right belongs to the next fine block
      const int64_t en = (J + 1) * rho;
      if (en < ne) acc += a.tout.lp[en * mc + c] * nxt[(xb + rho) * M];
      a.rc_out[J * mc + c] = acc;
    }
  }
}

}  // namespace aggmg
