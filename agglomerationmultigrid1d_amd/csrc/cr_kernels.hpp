// Coarsest-level direct solve of libaggmg_hip: block cyclic reduction (gfx950, wave64, fp64),
// factored once on the device (setup.hip), solved per cycle here.  Replaces `A_n \ rhs_n`
// (UMFPACK) of src/solvers.jl:39.
//
// Level l holds n block rows  a_i x_{i-1} + b_i x_i + c_i x_{i+1} = d_i  (m x m blocks).
// Odd rows are eliminated with their pivoted LU factors (never an explicit inverse):
//   forward   d'_j     = d_{2j} - a_{2j} (b_{2j-1} \ d_{2j-1}) - c_{2j} (b_{2j+1} \ d_{2j+1})
//   backward  x_{2j+1} = b_{2j+1} \ (d_{2j+1} - a_{2j+1} x_{2j} - c_{2j+1} x_{2j+2})
//
// Schedule.  A dependent chain of log2(n) levels is latency, not bandwidth, so the levels are
// taken three at a time: a *thread* owns the 2^3 + 1 blocks [8b, 8b + 8] of its sub-chunk, keeps
// them in registers and runs three levels on them without any synchronisation -- every factor it
// needs is addressable up front, i.e. one memory round trip per step instead of one per level.
// Sub-chunk boundary blocks stay even on all three levels; each of the two sub-chunks sharing one
// adds its own side's terms (R: the right-hand sub-chunk incl. d itself, L: the left-hand one) and
// the consumer sums the two.  A workgroup repeats such steps on the boundary blocks in LDS until
// its chunk of 2^q blocks is down to its two end blocks (a "stage": one launch forward, one
// backward; the first step streams the level's vector and factors from HBM, all later ones are
// 8, 64, ... times smaller); the boundary system of all chunks is the next stage's input, and
// what fits one workgroup (the "tail") is reduced to a single block and solved back in the same
// launch -- by the workgroup of the last stage that finishes last.  2^24 scalar rows: one stage
// (q = 12) + tail = 2 launches forward/tail, 1 backward.
//
// The forward pass touches the off-diagonal blocks of the even rows only, the backward pass those
// of the odd rows: stored apart (parity-split, a and c of one row adjacent) so that every fetched
// line is used in full.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aggmg {

constexpr int kCrThreads = 256;
// (tuning hook: __attribute__((amdgpu_waves_per_eu(1, 2))) lets the compiler spend the register file on loads in
// flight -- measured: -5 % at block size 2 with 256 workgroups, +25 % on the 4096-workgroup scalar system; off)
#ifndef CR_WAVES_ATTR
#define CR_WAVES_ATTR
#endif
constexpr int kCrTailRows = 4096;  // scalar rows (blocks * m) the single-workgroup tail takes
constexpr int kCrMaxLevels = 40;
constexpr int kCrMaxStageLevels = 12;
constexpr int kCrMaxSteps = 8;

struct CrLevel {
  const double* fe;    // [n_even][2][M][M]  (a_{2j}, c_{2j})
  const double* fo;    // [n_odd][2][M][M]   (a_{2j+1}, c_{2j+1})
  const double* lu;    // [n_odd][M][M]  unit-lower L and U of the row-permuted b_{2j+1}
  const int32_t* perm; // [n_odd][M]     row permutation: (P b) = L U, solve uses rhs[perm[k]]
  int64_t n, n_even, n_odd;
};

// levels a thread takes per step: register budget (2^Q + 1 + ... blocks of M doubles)
template <int M>
struct CrRadix {
  static constexpr int Q = M <= 4 ? 3 : 2;
};

// One launch: q levels in steps; step s runs local levels [step_a[s], step_a[s + 1]).
// LDS (doubles): for s = 1 .. nsteps the boundary vectors of local level step_a[s]:
//   R at lds_off[s], L at lds_off[s] + cnt_s, X (back substitution) at lds_xoff[s];
//   cnt_s = ((1 << (q - step_a[s])) + 1) * M.
struct CrStageArgs {
  CrLevel lv[kCrMaxStageLevels];
  int q;
  int nsteps;
  int step_a[kCrMaxSteps + 1];
  int lds_off[kCrMaxSteps + 1];
  int lds_xoff[kCrMaxSteps + 1];
  int lds_total;
  int64_t n_out;  // blocks left after the q levels
  int64_t c0;     // first chunk of this launch (element-partitioned runs launch a sub-range)
  double* stack;  // [n_chunks][stack_stride]: summed inputs of steps 1 .. nsteps-1, or null
  int stack_stride;
  int tail;       // one chunk = the whole system, n_out == 1: the last block is solved with lu_last
  int img;        // step 0 through the wave's LDS image where a wave's 64 sub-chunks are interior (M <= 2, see CrImg)
  double* stack0; // [3 M][stack0_stride]: the odd blocks of sub-levels 1, 2 of every step-0 sub-chunk (forward ->
  int64_t stack0_stride;  //              backward: no recomputation there), or null
  int dstride;    // doubles between consecutive blocks of the step-0 input vectors d0 / d0b (0: M, contiguous)
  int ostride;    // same for the boundary rows a forward stage writes (partR / partL); 2 M = interleaved per chunk
  const double* lu_last;
  const int32_t* perm_last;
};

__device__ __forceinline__ int64_t cr_level_n(const CrStageArgs& A, int l) { return l < A.q ? A.lv[l].n : A.n_out; }

// register-resident sub-chunk: sub-level i holds 2^(QS-i) + 1 blocks, stacked
template <int QS, int I>
struct CrOff {
  static constexpr int blocks = CrOff<QS, I - 1>::blocks + (1 << (QS - (I - 1))) + 1;
};
template <int QS>
struct CrOff<QS, 0> {
  static constexpr int blocks = 0;
};

// (a, c) of one row: 2 M^2 doubles, 16-byte aligned
template <int M>
__device__ __forceinline__ void cr_load_pair(const double* __restrict__ p, double (&ac)[2 * M * M]) {
  const double2* p2 = reinterpret_cast<const double2*>(p);
#pragma unroll
  for (int k = 0; k < M * M; ++k) {
    const double2 t = p2[k];
    ac[2 * k] = t.x;
    ac[2 * k + 1] = t.y;
  }
}

// y = b \ r for one block, r in registers: rows gathered through the stored permutation by select
// chains, then unit-lower and upper substitution (the getrs order)
template <int M>
__device__ __forceinline__ void cr_lu_solve_reg(const double* __restrict__ lu, const int32_t* __restrict__ perm,
                                                const double* r, double (&y)[M]) {
  if constexpr (M == 1) {
    y[0] = r[0] / lu[0];
  } else {
    double f[M * M];
#pragma unroll
    for (int k = 0; k < M * M; ++k) f[k] = lu[k];
    // (opaque copies: left as loads, the selects are folded into a variable-offset access and the
    // whole sub-chunk array drops out of registers into scratch)
    double rv[M];
#pragma unroll
    for (int q = 0; q < M; ++q) {
      rv[q] = r[q];
      asm("" : "+v"(rv[q]));
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const int32_t pk = perm[k];
      double v = rv[0];
#pragma unroll
      for (int q = 1; q < M; ++q) v = (pk == q) ? rv[q] : v;
      y[k] = v;
    }
#pragma unroll
    for (int i = 1; i < M; ++i) {
      double s = y[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= f[i * M + j] * y[j];
      y[i] = s;
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      double s = y[i];
#pragma unroll
      for (int j = i + 1; j < M; ++j) s -= f[i * M + j] * y[j];
      y[i] = s / f[i * M + i];
    }
  }
}

// The sub-chunk code below is branch-free on purpose: a load under a run-time condition cannot be
// hoisted, and every sub-level would wait a full memory round trip of its own -- the very latency
// chain this schedule is there to remove.  Rows past the end of the level are handled by clamping
// their index to a row that exists (loads stay in bounds) and discarding the value with a select.

// sub-levels I .. QS-1 of sub-chunk b forward: v[sub-level I] -> v[sub-level I + 1]
template <int M, int QS, int I>
__device__ __forceinline__ void cr_loc_fwd(const CrLevel* lv, int64_t b, double* v) {
  if constexpr (I < QS) {
    const CrLevel& L = lv[I];
    constexpr int NB1 = 1 << (QS - I - 1);  // next sub-level: blocks 0 .. NB1
    const int64_t tlo1 = b << (QS - I - 1);
    int64_t thi = (b + 1) << (QS - I);
    if (thi > L.n - 1) thi = L.n - 1;
    const int64_t tlo = tlo1 << 1;
    double* d = v + CrOff<QS, I>::blocks * M;
    double* dn = v + CrOff<QS, I + 1>::blocks * M;
    // b \ d of the odd rows, each used by both even neighbours; zero for rows that do not exist
    double y[NB1][M];
#pragma unroll
    for (int jj = 0; jj < NB1; ++jj) {
      const bool ok = tlo + 2 * jj + 1 <= thi;
      int64_t idx = tlo1 + jj;
      if (idx > L.n_odd - 1) idx = L.n_odd - 1;
      double t[M];
      cr_lu_solve_reg<M>(L.lu + idx * (M * M), L.perm + idx * M, d + (2 * jj + 1) * M, t);
#pragma unroll
      for (int e = 0; e < M; ++e) y[jj][e] = ok ? t[e] : 0.0;
    }
#pragma unroll
    for (int jj = 0; jj <= NB1; ++jj) {
      int64_t j = tlo1 + jj;
      if (j > L.n_even - 1) j = L.n_even - 1;  // a row that does not exist: its result is never read
      double ac[2 * M * M], acc[M];
      cr_load_pair<M>(L.fe + j * (2 * M * M), ac);
#pragma unroll
      for (int e = 0; e < M; ++e) acc[e] = d[(2 * jj) * M + e];
      if (jj > 0) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int k = 0; k < M; ++k) acc[i] -= ac[i * M + k] * y[jj > 0 ? jj - 1 : 0][k];
      }
      if (jj < NB1) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int k = 0; k < M; ++k) acc[i] -= ac[M * M + i * M + k] * y[jj < NB1 ? jj : 0][k];
      }
#pragma unroll
      for (int e = 0; e < M; ++e) dn[jj * M + e] = acc[e];
    }
    cr_loc_fwd<M, QS, I + 1>(lv, b, v);
  }
}

// sub-levels I .. 0 of sub-chunk b backward: x of sub-level I + 1 (in v) -> x of sub-level I,
// overwriting the reduced right-hand sides in place (odd rows read only their own d)
template <int M, int QS, int I>
__device__ __forceinline__ void cr_loc_bwd(const CrLevel* lv, int64_t b, double* v) {
  if constexpr (I >= 0) {
    const CrLevel& L = lv[I];
    constexpr int NB1 = 1 << (QS - I - 1);
    const int64_t tlo1 = b << (QS - I - 1);
    const int64_t tlo = tlo1 << 1;
    double* d = v + CrOff<QS, I>::blocks * M;
    const double* xn = v + CrOff<QS, I + 1>::blocks * M;
#pragma unroll
    for (int jj = 0; jj < NB1; ++jj) {  // odd rows 2 jj + 1 (rows past the end: values never written out)
      const int64_t r = tlo + 2 * jj + 1;
      int64_t idx = tlo1 + jj;
      if (idx > L.n_odd - 1) idx = L.n_odd - 1;
      double ac[2 * M * M], rhs[M], x[M];
      cr_load_pair<M>(L.fo + idx * (2 * M * M), ac);
      const bool has_next = r + 1 < L.n;
      double xr[M];
#pragma unroll
      for (int k = 0; k < M; ++k) xr[k] = has_next ? xn[(jj + 1) * M + k] : 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double s = d[(2 * jj + 1) * M + i];
#pragma unroll
        for (int k = 0; k < M; ++k) s -= ac[i * M + k] * xn[jj * M + k];
#pragma unroll
        for (int k = 0; k < M; ++k) s -= ac[M * M + i * M + k] * xr[k];
        rhs[i] = s;
      }
      cr_lu_solve_reg<M>(L.lu + idx * (M * M), L.perm + idx * M, rhs, x);
#pragma unroll
      for (int e = 0; e < M; ++e) d[(2 * jj + 1) * M + e] = x[e];
    }
#pragma unroll
    for (int jj = 0; jj <= NB1; ++jj)  // even rows: the coarser solution
#pragma unroll
      for (int e = 0; e < M; ++e) d[(2 * jj) * M + e] = xn[jj * M + e];
    cr_loc_bwd<M, QS, I - 1>(lv, b, v);
  }
}


// ---------------------------------------------------------------------------------------------
// Step 0 of a stage through LDS (block sizes 1 and 2, three levels per step).
//
// A lane that keeps its sub-chunk's ~150 factor values in registers can have only a fraction of their loads
// in flight: measured (r02) a stage launch issues them in about a dozen dependent batches, one HBM round
// trip each, with one wave per SIMD to hide them behind -- 2.7 TB/s.  Here a WAVE stages everything its 64
// sub-chunks need with LDS-DMA loads (global_load_lds_dwordx4: no register destination, so all ~55 KiB are in
// flight at once), waits once, and the lanes then read their values from LDS when the arithmetic wants them.
// The DMA writes lane-linear (destination = wave-uniform base + lane * 16), the SOURCE address is per lane:
// piece k (16 bytes) of every lane's contiguous run goes to region + k * 1024 + lane * 16, so that a lane's reads
// of one piece hit 64 consecutive 16-byte slots -- conflict-free ds_read_b128 -- while each lane still streams
// whole cache lines from its own part of the arrays.
// Used where all 64 sub-chunks of the wave are interior (no clamping, no ragged end); the few others keep the
// register path above.  Groups of 64 sub-chunks are aligned globally (chunks hold multiples of 64 of them), so
// which path a sub-chunk takes does not depend on the chunk size or the launch.
// ---------------------------------------------------------------------------------------------
// bytes of a region whose lanes each own `bytes` contiguous bytes (8: two lanes share a 16-byte piece; the DMA
// still writes a full KiB, see cr_glds)
constexpr int cr_img_reg(int bytes) { return bytes >= 16 ? (bytes / 16) * 1024 : 1024; }
constexpr int cr_img_nb1(int I) { return 4 >> I; }
template <int M> constexpr int cr_img_oF(int I) { return I == 0 ? cr_img_reg(8 * M * 8) : cr_img_oF<M>(I - 1) + cr_img_reg(cr_img_nb1(I - 1) * 2 * M * M * 8); }
// (row 0 of the sub-chunk behind the wave's last one, levels 0 .. 2: one DMA instruction, M^2 pieces per level)
template <int M> constexpr int cr_img_oFX(int I) { return cr_img_oF<M>(3) + I * 16 * M * M; }
template <int M> constexpr int cr_img_oU(int I) { return I == 0 ? cr_img_oF<M>(3) + 1024 : cr_img_oU<M>(I - 1) + cr_img_reg(cr_img_nb1(I - 1) * M * M * 8); }
template <int M> constexpr int cr_img_oP(int I) { return I == 0 ? cr_img_oU<M>(3) : cr_img_oP<M>(I - 1) + (M > 1 ? cr_img_reg(cr_img_nb1(I - 1) * M * 4) : 0); }

template <int M>
struct CrImg {
  static_assert(M == 1 || M == 2, "LDS image: block sizes 1 and 2");
  static constexpr int kRow = 2 * M * M * 8;  // bytes of an (a, c) row of fe / fo
  static constexpr int kLu = M * M * 8;       // bytes of an lu row
  static constexpr int kPm = M * 4;           // bytes of a perm row (not staged for M = 1: no permutation)
  static constexpr int kDb = 8 * M * 8;       // bytes of a sub-chunk's 8 blocks of d
  static constexpr int nb1(int I) { return cr_img_nb1(I); }
  static constexpr int oD = 0;                                             // d: 8 blocks per lane
  static constexpr int oF(int I) { return cr_img_oF<M>(I); }               // fe / fo rows of level I
  static constexpr int oFX(int I) { return cr_img_oFX<M>(I); }             // row 0 of the sub-chunk after the wave's last
  static constexpr int oU(int I) { return cr_img_oU<M>(I); }               // lu rows
  static constexpr int oP(int I) { return cr_img_oP<M>(I); }               // perm rows
  static constexpr int kBytes = (cr_img_oP<M>(3) + 1023) & ~1023;

  // address of byte `off` of the lane's run in a region of `bytes`-per-lane
  template <int BYTES>
  static __device__ __forceinline__ const char* at(const char* region, int lane, int off) {
    if constexpr (BYTES >= 16)
      return region + (off >> 4) * 1024 + lane * 16 + (off & 15);
    else
      return region + lane * 8 + off;
  }
};

// lane's `BYTES` contiguous bytes at gbase + (first + lane) * BYTES -> the region (see CrImg::at).
// EVERY lane takes part in every DMA instruction: under a partial exec mask the wave's slots came out wrong on
// gfx950 (measured: the compiler had split the wave around an `if (lane < n)` and issued the wave-wide loads
// once per part) -- lanes with nothing of their own to fetch repeat another lane's piece.
template <int BYTES>
__device__ __forceinline__ void cr_glds(const void* gbase, int64_t first, int lane, char* region) {
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  if constexpr (BYTES >= 16) {
    const char* src = static_cast<const char*>(gbase) + (first + lane) * BYTES;
#pragma unroll
    for (int k = 0; k < BYTES / 16; ++k)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + k * 16), (lptr_t)(region + k * 1024), 16, 0, 0);
  } else {
    static_assert(BYTES == 8, "runs of 8 bytes or multiples of 16");
    // lanes 2 l, 2 l + 1 share piece l (first is even: 16-byte aligned source); lanes 32 .. 63 repeat 0 .. 31
    const char* src = static_cast<const char*>(gbase) + (first + 2 * (lane & 31)) * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)region, 16, 0, 0);
  }
}

// row 0 of sub-chunk b0 + 64 at levels 0 .. 2 (row NB1 of lane 63): lane l < 3 M^2 fetches piece l % M^2 of level
// l / M^2, the others repeat lane 0's
template <int M, bool FWD>
__device__ __forceinline__ void cr_stage_extra(const CrLevel* lv, int64_t b0, int lane, char* img) {
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  using G = CrImg<M>;
  const int l = lane < 3 * M * M ? lane : 0;
  const int I = l / (M * M), t = l % (M * M);
  const double* rows = I == 0 ? (FWD ? lv[0].fe : lv[0].fo) : I == 1 ? (FWD ? lv[1].fe : lv[1].fo) : (FWD ? lv[2].fe : lv[2].fo);
  const char* src = reinterpret_cast<const char*>(rows) + (b0 + 64) * ((4 >> I) * G::kRow) + t * 16;
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(img + G::oFX(0)), 16, 0, 0);
}

// the factors of levels 0 .. 2 for the wave's sub-chunks b0 .. b0 + 63 (forward: fe; backward: fo), lu, perm
template <int M, bool FWD, int I>
__device__ __forceinline__ void cr_stage_levels(const CrLevel* lv, int64_t b0, int lane, char* img) {
  if constexpr (I < 3) {
    using G = CrImg<M>;
    constexpr int NB1 = G::nb1(I);
    const CrLevel& L = lv[I];
    const double* rows = FWD ? L.fe : L.fo;
    cr_glds<NB1 * G::kRow>(rows, b0, lane, img + G::oF(I));
    if constexpr (FWD && I == 0) cr_stage_extra<M, FWD>(lv, b0, lane, img);
    cr_glds<NB1 * G::kLu>(L.lu, b0, lane, img + G::oU(I));
    if constexpr (M > 1) cr_glds<NB1 * G::kPm>(L.perm, b0, lane, img + G::oP(I));
    cr_stage_levels<M, FWD, I + 1>(lv, b0, lane, img);
  }
}

template <int M, int I>
__device__ __forceinline__ void cr_img_lu(const char* img, int lane, int jj, double (&f)[M * M], int32_t (&pm)[M]) {
  using G = CrImg<M>;
  constexpr int NB1 = G::nb1(I);
  if constexpr (M == 1) {
    f[0] = *reinterpret_cast<const double*>(G::template at<NB1 * G::kLu>(img + G::oU(I), lane, jj * G::kLu));
    pm[0] = 0;
  } else {
#pragma unroll
    for (int t = 0; t < M * M / 2; ++t) {
      const double2 w = *reinterpret_cast<const double2*>(G::template at<NB1 * G::kLu>(img + G::oU(I), lane, jj * G::kLu + t * 16));
      f[2 * t] = w.x;
      f[2 * t + 1] = w.y;
    }
    const int2 q = *reinterpret_cast<const int2*>(G::template at<NB1 * G::kPm>(img + G::oP(I), lane, jj * G::kPm));
    pm[0] = q.x;
    pm[1] = q.y;
  }
}

// (a, c) of row jj of the lane's sub-chunk at level I; row NB1 is row 0 of the next sub-chunk (next lane, or the
// extra pieces for lane 63)
template <int M, int I>
__device__ __forceinline__ void cr_img_row(const char* img, int lane, int jj, double (&ac)[2 * M * M]) {
  using G = CrImg<M>;
  constexpr int NB1 = G::nb1(I);
  const bool next = jj == NB1;
  const bool extra = next && lane == 63;
  const char* p = extra ? img + G::oFX(I) : G::template at<NB1 * G::kRow>(img + G::oF(I), next ? lane + 1 : lane, next ? 0 : jj * G::kRow);
  const int stride = extra ? 16 : 1024;
#pragma unroll
  for (int t = 0; t < M * M; ++t) {
    const double2 w = *reinterpret_cast<const double2*>(p + t * stride);
    ac[2 * t] = w.x;
    ac[2 * t + 1] = w.y;
  }
}

// y = b \ r with the factors already in registers (the operation order of cr_lu_solve_reg)
template <int M>
__device__ __forceinline__ void cr_lu_solve_vals(const double (&f)[M * M], const int32_t (&pm)[M], const double* r, double (&y)[M]) {
  if constexpr (M == 1) {
    y[0] = r[0] / f[0];
  } else {
    double rv[M];
#pragma unroll
    for (int q = 0; q < M; ++q) {
      rv[q] = r[q];
      asm("" : "+v"(rv[q]));
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {
      double v = rv[0];
#pragma unroll
      for (int q = 1; q < M; ++q) v = (pm[k] == q) ? rv[q] : v;
      y[k] = v;
    }
#pragma unroll
    for (int i = 1; i < M; ++i) {
      double s = y[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= f[i * M + j] * y[j];
      y[i] = s;
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      double s = y[i];
#pragma unroll
      for (int j = i + 1; j < M; ++j) s -= f[i * M + j] * y[j];
      y[i] = s / f[i * M + i];
    }
  }
}

// cr_loc_fwd for an interior sub-chunk, factors from the image
template <int M, int I>
__device__ __forceinline__ void cr_fwd_img(const char* img, int lane, double* v) {
  if constexpr (I < 3) {
    constexpr int NB1 = 1 << (3 - I - 1);
    double* d = v + CrOff<3, I>::blocks * M;
    double* dn = v + CrOff<3, I + 1>::blocks * M;
    double y[NB1][M];
#pragma unroll
    for (int jj = 0; jj < NB1; ++jj) {
      double f[M * M];
      int32_t pm[M];
      cr_img_lu<M, I>(img, lane, jj, f, pm);
      cr_lu_solve_vals<M>(f, pm, d + (2 * jj + 1) * M, y[jj]);
    }
#pragma unroll
    for (int jj = 0; jj <= NB1; ++jj) {
      double ac[2 * M * M], acc[M];
      cr_img_row<M, I>(img, lane, jj, ac);
#pragma unroll
      for (int e = 0; e < M; ++e) acc[e] = d[(2 * jj) * M + e];
      if (jj > 0) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int k = 0; k < M; ++k) acc[i] -= ac[i * M + k] * y[jj > 0 ? jj - 1 : 0][k];
      }
      if (jj < NB1) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int k = 0; k < M; ++k) acc[i] -= ac[M * M + i * M + k] * y[jj < NB1 ? jj : 0][k];
      }
#pragma unroll
      for (int e = 0; e < M; ++e) dn[jj * M + e] = acc[e];
    }
    cr_fwd_img<M, I + 1>(img, lane, v);
  }
}

// cr_loc_bwd for an interior sub-chunk (every row has a right neighbour), factors from the image
template <int M, int I>
__device__ __forceinline__ void cr_bwd_img(const char* img, int lane, double* v) {
  if constexpr (I >= 0) {
    constexpr int NB1 = 1 << (3 - I - 1);
    double* d = v + CrOff<3, I>::blocks * M;
    const double* xn = v + CrOff<3, I + 1>::blocks * M;
#pragma unroll
    for (int jj = 0; jj < NB1; ++jj) {
      double ac[2 * M * M], rhs[M], x[M], f[M * M];
      int32_t pm[M];
      cr_img_row<M, I>(img, lane, jj, ac);
      cr_img_lu<M, I>(img, lane, jj, f, pm);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double s = d[(2 * jj + 1) * M + i];
#pragma unroll
        for (int k = 0; k < M; ++k) s -= ac[i * M + k] * xn[jj * M + k];
#pragma unroll
        for (int k = 0; k < M; ++k) s -= ac[M * M + i * M + k] * xn[(jj + 1) * M + k];
        rhs[i] = s;
      }
      cr_lu_solve_vals<M>(f, pm, rhs, x);
#pragma unroll
      for (int e = 0; e < M; ++e) d[(2 * jj + 1) * M + e] = x[e];
    }
#pragma unroll
    for (int jj = 0; jj <= NB1; ++jj)
#pragma unroll
      for (int e = 0; e < M; ++e) d[(2 * jj) * M + e] = xn[jj * M + e];
    cr_bwd_img<M, I - 1>(img, lane, v);
  }
}

// the lane's 8 blocks of d from the image into sub-level 0 of v
template <int M>
__device__ __forceinline__ void cr_img_d(const char* img, int lane, double* v) {
  using G = CrImg<M>;
#pragma unroll
  for (int k = 0; k < 8 * M / 2; ++k) {
    const double2 t = *reinterpret_cast<const double2*>(G::template at<G::kDb>(img + G::oD, lane, k * 16));
    v[2 * k] = t.x;
    v[2 * k + 1] = t.y;
  }
}

// geometry of step s for chunk c
struct CrStepGeom {
  int a, a1;            // local levels in / out
  int64_t lo_in, lo_out;
  int64_t n_in, n_out;  // blocks of the two levels (whole system)
  int nb;               // sub-chunks of this workgroup
};

template <int QS>
__device__ __forceinline__ CrStepGeom cr_step_geom(const CrStageArgs& A, int s, int64_t c, bool wg_shared) {
  CrStepGeom g;
  g.a = A.step_a[s];
  g.a1 = g.a + QS;
  g.lo_in = c << (A.q - g.a);
  g.lo_out = c << (A.q - g.a1);
  g.n_in = cr_level_n(A, g.a);
  g.n_out = cr_level_n(A, g.a1);
  int64_t hi_out = (c + 1) << (A.q - g.a1);
  if (hi_out > g.n_out - 1) hi_out = g.n_out - 1;
  g.nb = (int)(hi_out - g.lo_out + 1) - (wg_shared ? 1 : 0);
  return g;
}

// sub-level 0 of sub-chunk b from the step's input vector: global d0 (+ d0b) for step 0, the LDS
// boundary vectors R + L otherwise.  A shared right boundary's value belongs to the next sub-chunk
// (which reads R + L in full) -- except the workgroup's own right end in the later steps
// (carry_rb): no sub-chunk of this workgroup owns it, and the L terms the earlier steps collected
// for it have to travel on to partL.
template <int M, int QS>
__device__ __forceinline__ void cr_loc_load(const CrStageArgs& A, int s, const CrStepGeom& g, int64_t b,
                                            const double* __restrict__ d0, const double* __restrict__ d0b,
                                            const double* sh, double* v, int64_t& thi, bool& tshared,
                                            bool carry_rb) {
  constexpr int NB = 1 << QS;
  const int64_t tlo = b << QS;
  thi = (b + 1) << QS;
  tshared = thi <= g.n_in - 1;
  if (!tshared) thi = g.n_in - 1;
  if (s == 0) {
    const int ds = A.dstride ? A.dstride : M;
    if (tshared && ds == M) {  // a full sub-chunk: NB * M contiguous doubles, 16-byte aligned
      const double2* p = reinterpret_cast<const double2*>(d0 + tlo * M);
      if (d0b) {
        const double2* pb = reinterpret_cast<const double2*>(d0b + tlo * M);
#pragma unroll
        for (int k = 0; k < NB * M / 2; ++k) {
          const double2 t = p[k], u = pb[k];
          v[2 * k] = t.x + u.x;
          v[2 * k + 1] = t.y + u.y;
        }
      } else {
#pragma unroll
        for (int k = 0; k < NB * M / 2; ++k) {
          const double2 t = p[k];
          v[2 * k] = t.x;
          v[2 * k + 1] = t.y;
        }
      }
#pragma unroll
      for (int e = 0; e < M; ++e) v[NB * M + e] = 0.0;
    } else {
      // (the loads of one vector in one batch: a test of d0b per element would serialise them)
#pragma unroll
      for (int k = 0; k <= NB; ++k) {
        const int64_t gk = tlo + k <= thi ? tlo + k : thi;
#pragma unroll
        for (int e = 0; e < M; ++e) v[k * M + e] = d0[gk * ds + e];
      }
      if (d0b) {
#pragma unroll
        for (int k = 0; k <= NB; ++k) {
          const int64_t gk = tlo + k <= thi ? tlo + k : thi;
#pragma unroll
          for (int e = 0; e < M; ++e) v[k * M + e] += d0b[gk * ds + e];
        }
      }
#pragma unroll
      for (int k = 0; k <= NB; ++k) {
        // (a shared right boundary belongs to the next sub-chunk -- strided inputs bring full sub-chunks here)
        const bool ok = tlo + k <= thi && !(tshared && k == NB);
#pragma unroll
        for (int e = 0; e < M; ++e) v[k * M + e] = ok ? v[k * M + e] : 0.0;
      }
    }
  } else {
    const int cnt = ((1 << (A.q - g.a)) + 1) * M;
    const double* R = sh + A.lds_off[s];
    const double* Lp = R + cnt;
#pragma unroll
    for (int k = 0; k <= NB; ++k) {
      const bool ok = tlo + k <= thi && !(tshared && k == NB && !carry_rb);
      const int idx = (int)((tlo + k <= thi ? tlo + k : thi) - g.lo_in) * M;
#pragma unroll
      for (int e = 0; e < M; ++e) {
        const double x = R[idx + e] + Lp[idx + e];
        v[k * M + e] = ok ? x : 0.0;
      }
    }
  }
}

// one sub-chunk of step s forward, factors from global memory into registers
template <int M, int QS>
__device__ __forceinline__ void cr_sub_forward(const CrStageArgs& A, int s, const CrStepGeom& g, int i, bool wg_shared,
                                               const double* __restrict__ d0, const double* __restrict__ d0b,
                                               double* sh, double* Rout, double* Lout) {
  const int64_t b = g.lo_out + i;
  double v[CrOff<QS, QS + 1>::blocks * M];
  int64_t thi;
  bool tshared;
  cr_loc_load<M, QS>(A, s, g, b, d0, d0b, sh, v, thi, tshared, wg_shared && s > 0 && i == g.nb - 1);
  cr_loc_fwd<M, QS, 0>(&A.lv[g.a], b, v);
  const double* top = v + CrOff<QS, QS>::blocks * M;
#pragma unroll
  for (int e = 0; e < M; ++e) Rout[i * M + e] = top[e];
  if (b + 1 <= g.n_out - 1) {
#pragma unroll
    for (int e = 0; e < M; ++e) Lout[(i + 1) * M + e] = top[M + e];
  }
}

template <int M, int QS>
__device__ __forceinline__ void cr_step_forward(const CrStageArgs& A, int s, int64_t c, bool wg_shared,
                                                const double* __restrict__ d0, const double* __restrict__ d0b,
                                                double* sh) {
  const CrStepGeom g = cr_step_geom<QS>(A, s, c, wg_shared);
  const int cnt_out = ((1 << (A.q - g.a1)) + 1) * M;
  double* Rout = sh + A.lds_off[s + 1];
  double* Lout = Rout + cnt_out;
  for (int i = threadIdx.x; i < g.nb; i += blockDim.x) cr_sub_forward<M, QS>(A, s, g, i, wg_shared, d0, d0b, sh, Rout, Lout);
}

// whether the 64 sub-chunks b0 .. b0 + 63 of a stage's first step (eight blocks each plus the shared right
// boundary) are all interior: every row the image stages exists, nothing is clamped
__device__ __forceinline__ bool cr_group_interior(const CrStageArgs& A, int64_t b0) { return ((b0 + 64) << 3) <= A.lv[0].n - 1; }

// Step 0 forward with the wave's LDS image (img: this wave's CrImg<M>::kBytes): interior groups of 64 sub-chunks
// stage + compute from LDS and leave the odd blocks of sub-levels 1 and 2 in stack0 for the back substitution;
// other groups take cr_sub_forward.
template <int M>
__device__ __forceinline__ void cr_step0_forward_img(const CrStageArgs& A, int64_t c, bool wg_shared,
                                                     const double* __restrict__ d0, double* sh, char* img) {
  constexpr int QS = 3;
  using G = CrImg<M>;
  const CrStepGeom g = cr_step_geom<QS>(A, 0, c, wg_shared);
  const int cnt_out = ((1 << (A.q - g.a1)) + 1) * M;
  double* Rout = sh + A.lds_off[1];
  double* Lout = Rout + cnt_out;
  const int lane = threadIdx.x & 63;
  for (int base = threadIdx.x & ~63; base < g.nb; base += blockDim.x) {   // (wave-uniform)
    const int i = base + lane;
    const int64_t b0 = g.lo_out + base;
    if (base + 64 <= g.nb && cr_group_interior(A, b0)) {
      cr_glds<G::kDb>(d0, b0, lane, img + G::oD);
      cr_stage_levels<M, true, 0>(&A.lv[0], b0, lane, img);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      double v[CrOff<QS, QS + 1>::blocks * M];
      cr_img_d<M>(img, lane, v);
#pragma unroll
      for (int e = 0; e < M; ++e) v[8 * M + e] = 0.0;   // the shared right boundary's value belongs to the next sub-chunk
      cr_fwd_img<M, 0>(img, lane, v);
      const double* top = v + CrOff<QS, QS>::blocks * M;
#pragma unroll
      for (int e = 0; e < M; ++e) Rout[i * M + e] = top[e];
#pragma unroll
      for (int e = 0; e < M; ++e) Lout[(i + 1) * M + e] = top[M + e];
      {
        const double* s1 = v + CrOff<QS, 1>::blocks * M;
        const double* s2 = v + CrOff<QS, 2>::blocks * M;
        double* st = A.stack0 + (b0 + lane);
#pragma unroll
        for (int e = 0; e < M; ++e) {
          st[(int64_t)e * A.stack0_stride] = s1[1 * M + e];
          st[(int64_t)(M + e) * A.stack0_stride] = s1[3 * M + e];
          st[(int64_t)(2 * M + e) * A.stack0_stride] = s2[1 * M + e];
        }
      }
      asm volatile("" ::: "memory");   // the next pass restages the image: keep its reads above
    } else if (i < g.nb) {
      cr_sub_forward<M, QS>(A, 0, g, i, wg_shared, d0, nullptr, sh, Rout, Lout);
    }
  }
}

// xtop: x of the step's output level, indexed by block - xtop_lo (LDS, or the global boundary
// solution); xout: x of the step's input level (LDS for s >= 1, the caller's vector for s == 0)
template <int M, int QS>
__device__ __forceinline__ void cr_sub_backward(const CrStageArgs& A, int s, const CrStepGeom& g, int i, bool wg_shared,
                                                const double* __restrict__ d0, const double* __restrict__ d0b,
                                                const double* xtop, int64_t xtop_lo, double* xout, int64_t xout_lo,
                                                double* sh) {
  constexpr int NB = 1 << QS;
  const int64_t b = g.lo_out + i;
  double v[CrOff<QS, QS + 1>::blocks * M];
  int64_t thi;
  bool tshared;
  cr_loc_load<M, QS>(A, s, g, b, d0, d0b, sh, v, thi, tshared, wg_shared && s > 0 && i == g.nb - 1);
  cr_loc_fwd<M, QS, 0>(&A.lv[g.a], b, v);
  double* top = v + CrOff<QS, QS>::blocks * M;
  const bool has_right = b + 1 <= g.n_out - 1;
  const int64_t br = has_right ? b + 1 : b;
#pragma unroll
  for (int e = 0; e < M; ++e) {
    top[e] = xtop[(b - xtop_lo) * M + e];
    const double xr = xtop[(br - xtop_lo) * M + e];
    top[M + e] = has_right ? xr : 0.0;
  }
  cr_loc_bwd<M, QS, QS - 1>(&A.lv[g.a], b, v);
  const int64_t tlo = b << QS;
  double* o = xout + (tlo - xout_lo) * M;
  if (tshared && s == 0) {
    double2* o2 = reinterpret_cast<double2*>(o);
#pragma unroll
    for (int k = 0; k < NB * M / 2; ++k) o2[k] = make_double2(v[2 * k], v[2 * k + 1]);
  } else {
    // own blocks; a shared right boundary is the next sub-chunk's, except that the last sub-chunk of
    // the workgroup leaves it in LDS for the finer steps
    const bool write_rb = tshared && s > 0 && i == g.nb - 1;
#pragma unroll
    for (int k = 0; k <= NB; ++k) {
      const bool w = k < NB ? (tlo + k <= thi) : (tshared ? write_rb : (tlo + k <= thi));
      if (w) {
#pragma unroll
        for (int e = 0; e < M; ++e) o[k * M + e] = v[k * M + e];
      }
    }
  }
}

template <int M, int QS>
__device__ __forceinline__ void cr_step_backward(const CrStageArgs& A, int s, int64_t c, bool wg_shared,
                                                 const double* __restrict__ d0, const double* __restrict__ d0b,
                                                 const double* xtop, int64_t xtop_lo, double* xout, int64_t xout_lo,
                                                 double* sh) {
  const CrStepGeom g = cr_step_geom<QS>(A, s, c, wg_shared);
  for (int i = threadIdx.x; i < g.nb; i += blockDim.x)
    cr_sub_backward<M, QS>(A, s, g, i, wg_shared, d0, d0b, xtop, xtop_lo, xout, xout_lo, sh);
}

// Step 0 backward with the wave's LDS image: interior groups read d, fo, lu, perm from LDS and the odd blocks of
// sub-levels 1, 2 from stack0 (written by cr_step0_forward_img: nothing is recomputed, fe is not read again)
template <int M>
__device__ __forceinline__ void cr_step0_backward_img(const CrStageArgs& A, int64_t c, bool wg_shared,
                                                      const double* __restrict__ d0, const double* xtop, int64_t xtop_lo,
                                                      double* xout, double* sh, char* img) {
  constexpr int QS = 3;
  using G = CrImg<M>;
  const CrStepGeom g = cr_step_geom<QS>(A, 0, c, wg_shared);
  const int lane = threadIdx.x & 63;
  for (int base = threadIdx.x & ~63; base < g.nb; base += blockDim.x) {
    const int i = base + lane;
    const int64_t b0 = g.lo_out + base;
    if (base + 64 <= g.nb && cr_group_interior(A, b0)) {
      const int64_t b = b0 + lane;
      cr_glds<G::kDb>(d0, b0, lane, img + G::oD);
      cr_stage_levels<M, false, 0>(&A.lv[0], b0, lane, img);
      double v[CrOff<QS, QS + 1>::blocks * M];
      {
        double* s1 = v + CrOff<QS, 1>::blocks * M;
        double* s2 = v + CrOff<QS, 2>::blocks * M;
        const double* st = A.stack0 + b;
#pragma unroll
        for (int e = 0; e < M; ++e) {
          s1[1 * M + e] = st[(int64_t)e * A.stack0_stride];
          s1[3 * M + e] = st[(int64_t)(M + e) * A.stack0_stride];
          s2[1 * M + e] = st[(int64_t)(2 * M + e) * A.stack0_stride];
        }
      }
      double* top = v + CrOff<QS, QS>::blocks * M;
#pragma unroll
      for (int e = 0; e < M; ++e) {
        top[e] = xtop[(b - xtop_lo) * M + e];
        top[M + e] = xtop[(b + 1 - xtop_lo) * M + e];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      cr_img_d<M>(img, lane, v);
      cr_bwd_img<M, QS - 1>(img, lane, v);
      double2* o2 = reinterpret_cast<double2*>(xout + (b << QS) * M);
#pragma unroll
      for (int k = 0; k < 8 * M / 2; ++k) o2[k] = make_double2(v[2 * k], v[2 * k + 1]);
      asm volatile("" ::: "memory");
    } else if (i < g.nb) {
      cr_sub_backward<M, QS>(A, 0, g, i, wg_shared, d0, nullptr, xtop, xtop_lo, xout, 0, sh);
    }
  }
}

// this wave's LDS image: behind the step vectors of the workgroup (host: cr_stage_lds_bytes)
template <int M>
__device__ __forceinline__ char* cr_wave_image(const CrStageArgs& A, double* sh) {
  if constexpr (M <= 2)
    return reinterpret_cast<char*>(sh) + (((size_t)A.lds_total * sizeof(double) + 1023) & ~(size_t)1023) + (threadIdx.x >> 6) * CrImg<M>::kBytes;
  else
    return nullptr;
}

template <int M>
__device__ __forceinline__ void cr_forward_steps(const CrStageArgs& A, int64_t c, bool wg_shared,
                                                 const double* __restrict__ d0, const double* __restrict__ d0b,
                                                 double* sh) {
  constexpr int Q = CrRadix<M>::Q;
  for (int s = 0; s < A.nsteps; ++s) {
    const int qs = A.step_a[s + 1] - A.step_a[s];
    if constexpr (M <= 2) {
      if (s == 0 && (A.img & 1) && qs == 3 && !d0b) {   // (A.img: the host sized the LDS for one image per wave)
        cr_step0_forward_img<M>(A, c, wg_shared, d0, sh, cr_wave_image<M>(A, sh));
        __syncthreads();
        continue;
      }
    }
    if (qs == 1) {
      cr_step_forward<M, 1>(A, s, c, wg_shared, d0, d0b, sh);
    } else if (qs == 2) {
      cr_step_forward<M, 2>(A, s, c, wg_shared, d0, d0b, sh);
    } else {
      if constexpr (Q >= 3) cr_step_forward<M, 3>(A, s, c, wg_shared, d0, d0b, sh);
    }
    __syncthreads();
  }
}

// xq: the solution at the chunk's end blocks (global, level-q block index), or null when the top
// vector is already in LDS (tail)
template <int M>
__device__ __forceinline__ void cr_backward_steps(const CrStageArgs& A, int64_t c, bool wg_shared,
                                                  const double* __restrict__ d0, const double* __restrict__ d0b,
                                                  const double* __restrict__ xq, double* __restrict__ x0, double* sh) {
  constexpr int Q = CrRadix<M>::Q;
  for (int s = A.nsteps - 1; s >= 0; --s) {
    const int a = A.step_a[s], a1 = A.step_a[s + 1];
    const int qs = a1 - a;
    const bool top_global = xq && s == A.nsteps - 1;
    const double* xtop = top_global ? xq : sh + A.lds_xoff[s + 1];
    const int64_t xtop_lo = top_global ? 0 : (c << (A.q - a1));
    double* xout = s ? sh + A.lds_xoff[s] : x0;
    const int64_t xout_lo = s ? (c << (A.q - a)) : 0;
    if constexpr (M <= 2) {
      if (s == 0 && (A.img & 2) && qs == 3 && !d0b) {
        cr_step0_backward_img<M>(A, c, wg_shared, d0, xtop, xtop_lo, xout, sh, cr_wave_image<M>(A, sh));
        __syncthreads();
        continue;
      }
    }
    if (qs == 1) {
      cr_step_backward<M, 1>(A, s, c, wg_shared, d0, d0b, xtop, xtop_lo, xout, xout_lo, sh);
    } else if (qs == 2) {
      cr_step_backward<M, 2>(A, s, c, wg_shared, d0, d0b, xtop, xtop_lo, xout, xout_lo, sh);
    } else {
      if constexpr (Q >= 3) cr_step_backward<M, 3>(A, s, c, wg_shared, d0, d0b, xtop, xtop_lo, xout, xout_lo, sh);
    }
    __syncthreads();
  }
}

// the calling workgroup solves the whole tail system (T.tail) for d0 (+ d0b) into x0
template <int M>
__device__ __forceinline__ void cr_tail_body(const CrStageArgs& T, const double* __restrict__ d0,
                                             const double* __restrict__ d0b, double* __restrict__ x0, double* sh) {
  for (int t = threadIdx.x; t < T.lds_total; t += blockDim.x) sh[t] = 0.0;
  __syncthreads();
  if (T.nsteps == 0) {  // a single block
    if (threadIdx.x == 0) {
      double r[M], y[M];
#pragma unroll
      for (int e = 0; e < M; ++e) r[e] = d0b ? d0[e] + d0b[e] : d0[e];
      cr_lu_solve_reg<M>(T.lu_last, T.perm_last, r, y);
#pragma unroll
      for (int e = 0; e < M; ++e) x0[e] = y[e];
    }
    return;
  }
  cr_forward_steps<M>(T, 0, false, d0, d0b, sh);
  if (threadIdx.x == 0) {
    const double* R = sh + T.lds_off[T.nsteps];
    double r[M], y[M];
#pragma unroll
    for (int e = 0; e < M; ++e) r[e] = R[e] + R[2 * M + e];
    cr_lu_solve_reg<M>(T.lu_last, T.perm_last, r, y);
    double* X = sh + T.lds_xoff[T.nsteps];
#pragma unroll
    for (int e = 0; e < M; ++e) X[e] = y[e];
  }
  __syncthreads();
  cr_backward_steps<M>(T, 0, false, d0, d0b, nullptr, x0, sh);
}

template <int M>
__global__ __launch_bounds__(kCrThreads) CR_WAVES_ATTR void cr_tail_kernel(CrStageArgs T, const double* __restrict__ d0,
                                                             const double* __restrict__ d0b,
                                                             double* __restrict__ x0) {
  extern __shared__ double sh[];
  cr_tail_body<M>(T, d0, d0b, x0, sh);
}

// One workgroup per chunk c = blocks [c 2^q, (c+1) 2^q] of the stage's first level: the chunk's
// end blocks go to partR[c] (this side's terms incl. d) / partL[c + 1] (terms for the next chunk's
// left end).  FUSE_TAIL: the workgroup that finishes last (ticket counter) goes on to solve the
// boundary system with the tail levels -- forward elimination and boundary solve in one launch.
template <int M, bool FUSE_TAIL>
__global__ __launch_bounds__(kCrThreads) CR_WAVES_ATTR void cr_stage_forward_kernel(CrStageArgs A, const double* __restrict__ d0,
                                                                      const double* __restrict__ d0b, double* partR,
                                                                      double* partL, CrStageArgs T, double* xq,
                                                                      unsigned int* ticket) {
  extern __shared__ double sh[];
  __shared__ unsigned int s_ticket;
  const int64_t c = A.c0 + blockIdx.x;
  const bool wg_shared = ((c + 1) << A.q) <= A.lv[0].n - 1;
  for (int t = threadIdx.x; t < A.lds_total; t += blockDim.x) sh[t] = 0.0;
  __syncthreads();
  cr_forward_steps<M>(A, c, wg_shared, d0, d0b, sh);
  {
    const double* R = sh + A.lds_off[A.nsteps];  // level q: blocks c, c + 1
    if (threadIdx.x < M) {
      const int os = A.ostride ? A.ostride : M;
      partR[c * os + threadIdx.x] = R[threadIdx.x];
      if (c + 1 <= A.n_out - 1) partL[(c + 1) * os + threadIdx.x] = R[2 * M + M + threadIdx.x];
    }
  }
  // the summed inputs of the later steps are kept for the back substitution (a few hundred values
  // per chunk) instead of recomputing every step there
  if (A.stack) {
    double* st = A.stack + c * (int64_t)A.stack_stride;
    int o = 0;
    for (int s = 1; s < A.nsteps; ++s) {
      const int cnt = ((1 << (A.q - A.step_a[s])) + 1) * M;
      const double* R = sh + A.lds_off[s];
      for (int t = threadIdx.x; t < cnt; t += blockDim.x) st[o + t] = R[t] + R[cnt + t];
      o += cnt;
    }
  }
  if constexpr (FUSE_TAIL) {
    __threadfence();  // partR / partL of this workgroup visible device-wide before the ticket
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    __threadfence();  // acquire: every other workgroup's partR / partL
    cr_tail_body<M>(T, partR, partL, xq, sh);
    if (threadIdx.x == 0) *ticket = 0u;  // ready for the next solve (stream-ordered)
  }
}

template <int M>
__global__ __launch_bounds__(kCrThreads) CR_WAVES_ATTR void cr_stage_backward_kernel(CrStageArgs A, const double* __restrict__ d0,
                                                                       const double* __restrict__ d0b,
                                                                       const double* __restrict__ xq,
                                                                       double* __restrict__ x0) {
  extern __shared__ double sh[];
  const int64_t c = A.c0 + blockIdx.x;
  const bool wg_shared = ((c + 1) << A.q) <= A.lv[0].n - 1;
  for (int t = threadIdx.x; t < A.lds_total; t += blockDim.x) sh[t] = 0.0;
  __syncthreads();
  if (A.nsteps > 1) {
    if (A.stack) {
      const double* st = A.stack + c * (int64_t)A.stack_stride;
      int o = 0;
      for (int s = 1; s < A.nsteps; ++s) {
        const int cnt = ((1 << (A.q - A.step_a[s])) + 1) * M;
        double* R = sh + A.lds_off[s];
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) R[t] = st[o + t];
        o += cnt;
      }
      __syncthreads();
    } else {
      cr_forward_steps<M>(A, c, wg_shared, d0, d0b, sh);
    }
  }
  cr_backward_steps<M>(A, c, wg_shared, d0, d0b, xq, x0, sh);
}

}  // namespace aggmg
