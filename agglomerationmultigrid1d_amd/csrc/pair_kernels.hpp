// Two levels of the V-cycle in one launch (src/solvers.jl:28-37 for levels k, k + 1 on the way down, :41-47 for
// k + 1, k on the way up): the small agglomerated levels of a hierarchy are a few tens of microseconds each -- at a
// rank's share of a partitioned run 9 - 21 us -- and a tile of level k + 1 is the restriction of a tile of level k, so
// the hand-over (the restricted residual going down, the coarse correction coming up) stays in LDS and the two
// launches become one.
//
// Levels taken: block-tridiagonal with DENSE off-diagonal blocks of size M (agglomerated DG levels, M = pAgg + 1 = 2),
// symmetric-packed block inverses (BtdLevel::bsym), two-mode transfers with one agglomeration ratio per level
// (TransferBtd::rho > 0, mc = 2).  The arithmetic of every element is that of btd_fused_kernel<M, false, ., true>
// -- same expressions in the same order -- so a cycle with paired launches equals the cycle with separate ones bit
// for bit (tests/test_gpu_pair.py); anything else (compressed off-diagonal blocks, agglomerates of different sizes,
// Gauss-Seidel sweeps, the preconditioned restriction) keeps the separate launches.
//
// Tile geometry, descent (halo h = nsweeps + 1 per level: nsweeps sweeps and the residual):
//   level b = k + 1:  te_b = own_b + 2 h elements at x_b = 0 .. te_b - 1, element Eb0 + x_b, Eb0 = tile * own_b - h
//   level a = k:      the children of all te_b elements plus h on both sides: te_a = te_b * rho_a + 2 h
// The residual of level a is valid on the children of the whole level-b tile, so the level-b right-hand side is
// complete in LDS (its owned part is also stored: the ascent reads it).  Ascent (halo nsweeps per level):
//   level a:  te_a = own_a + 2 nsweeps;   level b: the parents of that tile (cb elements) plus nsweeps on both sides.
#pragma once
#include "kernels.hpp"

namespace aggmg {

struct PairLevel {
  const double *bsym, *dblk, *sub, *sup;
  int64_t ne;
};

struct PairXfer {      // two coarse modes per element
  const double* lf;    // [N_f][2] rows of L
  const double* lf1;   // [N_f] their second entries when every first entry is exactly 1.0, else null
  int rho;
  int64_t nec;
};

struct PairArgs {
  PairLevel A, B;   // level a = k (finer), b = k + 1
  PairXfer ab, bc;  // a <-> b, b <-> b + 1
  double alpha;
  int nsweeps;
  // descent: rhs_a in; u_a, rhs_b, u_b, rhs_c out.  ascent: rhs_a, rhs_b, u_a (pre-smoothed), u_b (pre-smoothed),
  // uc (level b + 1) in; ua_out (and ub_out when not null) out
  const double *rhs_a, *rhs_b_in, *ua_in, *ub_in, *uc;
  double *u_a, *rhs_b, *u_b, *rhs_c, *ub_out;
  int own;    // descent: level-b elements owned per tile (multiple of bc.rho); ascent: level-a elements (multiple of ab.rho)
  int te_a, te_b;
  int hb;     // ascent: level-b elements beyond the parents of the owned level-a range on each side = ceil(nsweeps / rho_a)
  // ascent: tile of workgroup w = w + (w >= tile_split ? tile_skip : 0) -- a launch may cover the tiles at the two ends of
  // the level (the ones that read ghost values of the coarser level) or the ones in between (fused_tile_subset)
  int tile_split;
  int64_t tile_skip;
};

// one level's operator rows of a thread: row i of the packed symmetric inverse, rows i of Sup_{e-1} and Sup_e turned
// into rows of P = B^{-1} Sub and Q = B^{-1} Sup (Sub_e = Sup_{e-1}'), g = B^{-1} b -- btd_fused_kernel's DSYM path
template <int M>
__device__ __forceinline__ void pair_load_rows(const PairLevel& L, bool valid, int64_t e, int i, double bb, double (&Pr)[M],
                                               double (&Qr)[M], double& g) {
  double bi[M];
#pragma unroll
  for (int j = 0; j < M; ++j) bi[j] = 0.0;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    Pr[j] = 0.0;
    Qr[j] = 0.0;
  }
  if (valid) {
    constexpr int T = M * (M + 1) / 2;
    const int64_t row = e * M + i;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int lo_ = i < j ? i : j, hi_ = i < j ? j : i;
      bi[j] = L.bsym[e * T + lo_ * M - (lo_ * (lo_ - 1)) / 2 + (hi_ - lo_)];
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      Pr[j] = e > 0 ? L.sup[(row - M) * M + j] : 0.0;
      Qr[j] = L.sup[row * M + j];
    }
  }
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < M; ++j) acc += bi[j] * group_bcast<M>(bb, j);
  g = acc;
  double pn[M], qn[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    double pa = 0.0, qa = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      pa += bi[k] * group_bcast<M>(Pr[k], j);
      qa += bi[k] * group_bcast<M>(Qr[j], k);
    }
    pn[j] = pa;
    qn[j] = qa;
  }
#pragma unroll
  for (int j = 0; j < M; ++j) {
    Pr[j] = pn[j];
    Qr[j] = qn[j];
  }
}

// nsweeps block-Jacobi sweeps of a tile in LDS (ping-pong, one barrier per sweep); uu[] carries the thread's own rows
template <int M, int NS, int EPS>
__device__ __forceinline__ void pair_sweeps(int nsweeps, double alpha, int le, int i, const bool (&valid)[NS], const double (&g)[NS],
                                            const double (&Pr)[NS][M], const double (&Qr)[NS][M], double (&uu)[NS], double*& cur,
                                            double*& nxt) {
  for (int sw = 0; sw < nsweeps; ++sw) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int x = s * EPS + le;
      const double* um = cur + (x - 1) * M;
      const double* up = cur + (x + 1) * M;
      double acc = g[s];
#pragma unroll
      for (int j = 0; j < M; ++j) acc -= Pr[s][j] * um[j];
#pragma unroll
      for (int j = 0; j < M; ++j) acc -= Qr[s][j] * up[j];
      double un = uu[s] + alpha * (acc - uu[s]);
      if (!valid[s]) un = 0.0;
      uu[s] = un;
      nxt[x * M + i] = un;
    }
    __syncthreads();
    double* t = cur;
    cur = nxt;
    nxt = t;
  }
}

// explicit residual row r = b - A u with the operator's own entries, ascending column order (btd_fused_kernel's)
template <int M>
__device__ __forceinline__ double pair_residual_row(const PairLevel& L, int64_t row, double bb, const double* um, const double* ux,
                                                    const double* up) {
  double t = 0.0;
#pragma unroll
  for (int j = 0; j < M; ++j) t += L.sub[row * M + j] * um[j];
#pragma unroll
  for (int j = 0; j < M; ++j) t += L.dblk[row * M + j] * ux[j];
#pragma unroll
  for (int j = 0; j < M; ++j) t += L.sup[row * M + j] * up[j];
  return bb - t;
}

__device__ __forceinline__ void pair_l2(const PairXfer& X, int64_t row, double& lx, double& ly) {
  if (X.lf1) {  // unit first column: 1.0 * r is r, bit for bit what the stored 1.0 gives
    lx = 1.0;
    ly = AGGMG_LD(X.lf1[row]);
  } else {
    const double2 t2 = *reinterpret_cast<const double2*>(X.lf + row * 2);
    lx = t2.x;
    ly = t2.y;
  }
}

// LDS: [ two iterate buffers of (TEA + 2) * M doubles, each padded by one zero element on both sides | level-b vector
// of TEB * 2 doubles ]; the level-b iterate buffers reuse the front region
template <int M, int NSA, int NSB, int NT>
__global__ __launch_bounds__(NT) void btd_pair_down_kernel(PairArgs a) {
  constexpr int MB = 2;
  constexpr int EPSA = NT / M, TEA = EPSA * NSA;
  constexpr int EPSB = NT / MB, TEB = EPSB * NSB;
  static_assert((TEB + 2) * MB <= (TEA + 2) * M, "level-b buffers must fit the level-a ones");
  extern __shared__ double lds[];
  double* buf0 = lds + M;
  double* buf1 = lds + (TEA + 2) * M + M;
  double* rhsB = lds + 2 * (TEA + 2) * M;   // [TEB][2]

  const int tid = threadIdx.x;
  const int h = a.nsweeps + 1;
  const int64_t Eb0 = (int64_t)blockIdx.x * a.own - h;
  const int rhoA = a.ab.rho;
  const int64_t Ea0 = Eb0 * rhoA - h;

  // ---------------- level a: nsweeps sweeps from zero, residual, restriction into LDS ----------------------------
  {
    const int le = tid / M, i = tid - le * M;
    if (tid < M) {
      buf0[-M + tid] = 0.0;
      buf0[TEA * M + tid] = 0.0;
      buf1[-M + tid] = 0.0;
      buf1[TEA * M + tid] = 0.0;
    }
    double g[NSA], bb[NSA], uu[NSA], lx[NSA], ly[NSA];
    double Pr[NSA][M], Qr[NSA][M];
    bool valid[NSA];
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
      const int x = s * EPSA + le;
      const int64_t e = Ea0 + x;
      valid[s] = x < a.te_a && e >= 0 && e < a.A.ne;
      const int64_t row = e * M + i;
      bb[s] = valid[s] ? a.rhs_a[row] : 0.0;
      uu[s] = 0.0;                       // u = zeros below the finest level (src/solvers.jl:29-31)
      lx[s] = ly[s] = 0.0;
      if (valid[s]) pair_l2(a.ab, row, lx[s], ly[s]);
      pair_load_rows<M>(a.A, valid[s], e, i, bb[s], Pr[s], Qr[s], g[s]);
      buf0[x * M + i] = 0.0;
    }
    __syncthreads();
    double* cur = buf0;
    double* nxt = buf1;
    pair_sweeps<M, NSA, EPSA>(a.nsweeps, a.alpha, le, i, valid, g, Pr, Qr, uu, cur, nxt);
    // iterate of the children of the owned level-b elements
    const int xs0 = h + h * rhoA, xs1 = h + (h + a.own) * rhoA;
    double rr[NSA];
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
      const int x = s * EPSA + le;
      if (valid[s] && x >= xs0 && x < xs1) AGGMG_ST(a.u_a[(Ea0 + x) * M + i], uu[s]);
      rr[s] = 0.0;
      if (valid[s] && x >= h && x < a.te_a - h)
        rr[s] = pair_residual_row<M>(a.A, (Ea0 + x) * M + i, bb[s], cur + (x - 1) * M, cur + x * M, cur + (x + 1) * M);
    }
    __syncthreads();   // every residual has read the iterate: both buffers are free for the products
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
      const int x = s * EPSA + le;
      const bool in = valid[s] && x >= h && x < a.te_a - h;
      nxt[x * M + i] = in ? lx[s] * rr[s] : 0.0;
      cur[x * M + i] = in ? ly[s] * rr[s] : 0.0;
    }
    __syncthreads();
    // rhs_b = L' r for EVERY element of the level-b tile (ascending fine row, as the column dot of L')
    for (int t = tid; t < a.te_b * 2; t += NT) {
      const int Jl = t >> 1, c = t & 1;
      const int64_t J = Eb0 + Jl;
      double acc = 0.0;
      if (J >= 0 && J < a.B.ne) {
        const double* pr = (c ? cur : nxt) + (h + Jl * rhoA) * M;
        for (int k = 0; k < rhoA * M; ++k) acc += pr[k];
        if (Jl >= h && Jl < h + a.own) a.rhs_b[J * 2 + c] = acc;
      }
      rhsB[t] = acc;
    }
    __syncthreads();
  }

  // ---------------- level b: nsweeps sweeps from zero, residual, restriction to level b + 1 ----------------------
  {
    const int le = tid / MB, i = tid - le * MB;
    double* b0 = lds + MB;
    double* b1 = lds + (TEB + 2) * MB + MB;
    if (tid < MB) {
      b0[-MB + tid] = 0.0;
      b0[TEB * MB + tid] = 0.0;
      b1[-MB + tid] = 0.0;
      b1[TEB * MB + tid] = 0.0;
    }
    double g[NSB], bb[NSB], uu[NSB], lx[NSB], ly[NSB];
    double Pr[NSB][MB], Qr[NSB][MB];
    bool valid[NSB];
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
      const int x = s * EPSB + le;
      const int64_t e = Eb0 + x;
      valid[s] = x < a.te_b && e >= 0 && e < a.B.ne;
      const int64_t row = e * MB + i;
      bb[s] = valid[s] ? rhsB[x * MB + i] : 0.0;
      uu[s] = 0.0;
      lx[s] = ly[s] = 0.0;
      if (valid[s] && x >= h && x < h + a.own) pair_l2(a.bc, row, lx[s], ly[s]);
      pair_load_rows<MB>(a.B, valid[s], e, i, bb[s], Pr[s], Qr[s], g[s]);
      b0[x * MB + i] = 0.0;
    }
    __syncthreads();
    double* cur = b0;
    double* nxt = b1;
    pair_sweeps<MB, NSB, EPSB>(a.nsweeps, a.alpha, le, i, valid, g, Pr, Qr, uu, cur, nxt);
    double rr[NSB];
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
      const int x = s * EPSB + le;
      const bool own = valid[s] && x >= h && x < h + a.own;
      if (own) AGGMG_ST(a.u_b[(Eb0 + x) * MB + i], uu[s]);
      rr[s] = own ? pair_residual_row<MB>(a.B, (Eb0 + x) * MB + i, bb[s], cur + (x - 1) * MB, cur + x * MB, cur + (x + 1) * MB) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
      const int x = s * EPSB + le;
      const bool own = valid[s] && x >= h && x < h + a.own;
      nxt[x * MB + i] = own ? lx[s] * rr[s] : 0.0;
      cur[x * MB + i] = own ? ly[s] * rr[s] : 0.0;
    }
    __syncthreads();
    const int rhoB = a.bc.rho;
    const int ncoarse = a.own / rhoB;
    const int64_t J0 = ((int64_t)blockIdx.x * a.own) / rhoB;
    for (int t = tid; t < ncoarse * 2; t += NT) {
      const int Jl = t >> 1, c = t & 1;
      const int64_t J = J0 + Jl;
      if (J >= a.bc.nec) continue;
      const double* pr = (c ? cur : nxt) + (h + Jl * rhoB) * MB;
      double acc = 0.0;
      for (int k = 0; k < rhoB * MB; ++k) acc += pr[k];
      a.rhs_c[J * 2 + c] = acc;
    }
  }
}

// ascent: level b (prolongation from level b + 1, nsweeps sweeps) then level a (prolongation from the level-b tile in
// LDS, nsweeps sweeps)
template <int M, int NSA, int NSB, int NT>
__global__ __launch_bounds__(NT) void btd_pair_up_kernel(PairArgs a) {
  constexpr int MB = 2;
  constexpr int EPSA = NT / M, TEA = EPSA * NSA;
  constexpr int EPSB = NT / MB, TEB = EPSB * NSB;
  static_assert((TEB + 2) * MB <= (TEA + 2) * M, "level-b buffers must fit the level-a ones");
  extern __shared__ double lds[];
  double* uB = lds + 2 * (TEA + 2) * M;   // [TEB][2]: post-smoothed level-b iterate of the tile

  const int tid = threadIdx.x;
  const int ns = a.nsweeps;
  const int rhoA = a.ab.rho, rhoB = a.bc.rho;
  const int64_t tile = (int64_t)blockIdx.x + ((int)blockIdx.x >= a.tile_split ? a.tile_skip : 0);
  const int64_t Ea0 = tile * a.own - ns;
  const int64_t Eb0 = (tile * a.own) / rhoA - a.hb - ns;

  // ---------------- level b -------------------------------------------------------------------------------------
  {
    const int le = tid / MB, i = tid - le * MB;
    double* b0 = lds + MB;
    double* b1 = lds + (TEB + 2) * MB + MB;
    if (tid < MB) {
      b0[-MB + tid] = 0.0;
      b0[TEB * MB + tid] = 0.0;
      b1[-MB + tid] = 0.0;
      b1[TEB * MB + tid] = 0.0;
    }
    double g[NSB], bb[NSB], uu[NSB];
    double Pr[NSB][MB], Qr[NSB][MB];
    bool valid[NSB];
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
      const int x = s * EPSB + le;
      const int64_t e = Eb0 + x;
      valid[s] = x < a.te_b && e >= 0 && e < a.B.ne;
      const int64_t row = e * MB + i;
      bb[s] = 0.0;
      uu[s] = 0.0;
      if (valid[s]) {
        bb[s] = a.rhs_b_in[row];
        uu[s] = a.ub_in[row];
        // u += L uc : J = e / rho, ascending mode order (CSC scatter order)
        const int64_t J = e / rhoB;
        double l2x, l2y;
        pair_l2(a.bc, row, l2x, l2y);
        const double2 u2 = *reinterpret_cast<const double2*>(a.uc + J * 2);
        double add = l2x * u2.x;
        add += l2y * u2.y;
        uu[s] += add;
      }
      pair_load_rows<MB>(a.B, valid[s], e, i, bb[s], Pr[s], Qr[s], g[s]);
      b0[x * MB + i] = uu[s];
    }
    __syncthreads();
    double* cur = b0;
    double* nxt = b1;
    pair_sweeps<MB, NSB, EPSB>(ns, a.alpha, le, i, valid, g, Pr, Qr, uu, cur, nxt);
    const int64_t ob0 = (tile * a.own) / rhoA, ob1 = ob0 + a.own / rhoA;   // the parents of the owned level-a range
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
      const int x = s * EPSB + le;
      uB[x * MB + i] = uu[s];
      const int64_t e = Eb0 + x;
      if (a.ub_out && valid[s] && e >= ob0 && e < ob1) AGGMG_ST(a.ub_out[e * MB + i], uu[s]);
    }
    __syncthreads();
  }

  // ---------------- level a -------------------------------------------------------------------------------------
  {
    const int le = tid / M, i = tid - le * M;
    double* buf0 = lds + M;
    double* buf1 = lds + (TEA + 2) * M + M;
    if (tid < M) {
      buf0[-M + tid] = 0.0;
      buf0[TEA * M + tid] = 0.0;
      buf1[-M + tid] = 0.0;
      buf1[TEA * M + tid] = 0.0;
    }
    double g[NSA], bb[NSA], uu[NSA];
    double Pr[NSA][M], Qr[NSA][M];
    bool valid[NSA];
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
      const int x = s * EPSA + le;
      const int64_t e = Ea0 + x;
      valid[s] = x < a.te_a && e >= 0 && e < a.A.ne;
      const int64_t row = e * M + i;
      bb[s] = 0.0;
      uu[s] = 0.0;
      if (valid[s]) {
        bb[s] = a.rhs_a[row];
        uu[s] = a.ua_in[row];
        const int xb = (int)(e / rhoA - Eb0);
        double l2x, l2y;
        pair_l2(a.ab, row, l2x, l2y);
        const double2 u2 = *reinterpret_cast<const double2*>(uB + xb * 2);
        double add = l2x * u2.x;
        add += l2y * u2.y;
        uu[s] += add;
      }
      pair_load_rows<M>(a.A, valid[s], e, i, bb[s], Pr[s], Qr[s], g[s]);
    }
    // (the level-b buffers and this level's share the front of the LDS: all reads of uB -- behind them -- are done by
    // value above, the buffers themselves were last read in level b's sweeps, a barrier ago)
#pragma unroll
    for (int s = 0; s < NSA; ++s) buf0[(s * EPSA + le) * M + i] = uu[s];
    __syncthreads();
    double* cur = buf0;
    double* nxt = buf1;
    pair_sweeps<M, NSA, EPSA>(ns, a.alpha, le, i, valid, g, Pr, Qr, uu, cur, nxt);
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
      const int x = s * EPSA + le;
      if (valid[s] && x >= ns && x < ns + a.own) AGGMG_ST(a.u_a[(Ea0 + x) * M + i], uu[s]);
    }
  }
}

}  // namespace aggmg
