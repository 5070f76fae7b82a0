// Coarsest-level direct solve of libaggmg_hip: block cyclic reduction (gfx950, wave64, fp64),
// factored once on the device (setup.hip), solved per cycle here.  Replaces `A_n \ rhs_n`
// (UMFPACK) of src/solvers.jl:39.
//
// Level l holds n block rows  a_i x_{i-1} + b_i x_i + c_i x_{i+1} = d_i  (m x m blocks).
// Odd rows are eliminated by block elimination with pivoted LU factors of their diagonal blocks:
//   forward   d'_j     = d_{2j} - (a_{2j} b_{2j-1}^-1) d_{2j-1} - (c_{2j} b_{2j+1}^-1) d_{2j+1}
//             with the multipliers in brackets formed once at set-up from the factors (as the Schur complements are),
//   backward  x_{2j+1} = b_{2j+1} \ (d_{2j+1} - a_{2j+1} x_{2j} - c_{2j+1} x_{2j+2})   (triangular solves per cycle)
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
// what fits one workgroup (the "tail") is solved in a launch of its own: by parallel cyclic reduction
// (end of this file: block sizes 1 and 2, up to 1024 blocks) or reduced to a single block and solved
// back by the steps above.  2^24 scalar rows: one stage (q = 12) forward, tail, one stage backward
// = 3 launches.
//
// The forward pass touches the off-diagonal blocks of the even rows only, the backward pass those
// of the odd rows: stored apart (parity-split, a and c of one row adjacent) so that every fetched
// line is used in full.  The back substitution of a sub-chunk needs the reduced right-hand sides of the
// odd rows of its inner sub-levels (three blocks of the nine at three levels per step): the forward
// step stores them (`mid`) instead of the backward step reading the even rows' factors of all three
// sub-levels again to recompute them (r03: 0.22 -> 0.14 GB per backward launch at 2^20 blocks of 2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "host_plan.hpp"  // kCrTailRows, kCrMaxLevels, kCrMaxStageLevels, kCrMaxSteps and the step planner

namespace aggmg {

constexpr int kCrThreads = 256;
// (tuning hook: __attribute__((amdgpu_waves_per_eu(1, 2))) lets the compiler spend the register file on loads in
// flight -- measured: -5 % at block size 2 with 256 workgroups, +25 % on the 4096-workgroup scalar system; off)
#ifndef CR_WAVES_ATTR
#define CR_WAVES_ATTR
#endif

struct CrLevel {
  const double* fe;    // [n_even][2][M][M]  forward multipliers (a_{2j} b_{2j-1}^-1, c_{2j} b_{2j+1}^-1)
  const double* fo;    // [n_odd][2][M][M]   (a_{2j+1}, c_{2j+1})
  const double* lu;    // [n_odd][M][M]  unit-lower L and U of the row-permuted b_{2j+1}, diagonal of U as reciprocals
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
  double* mid;    // reduced right-hand sides of the odd rows of the inner sub-levels, written by the forward steps
  int64_t mid_off[kCrMaxSteps + 1];  // and read by the backward ones: step s, sub-chunk b at mid_off[s] + b (2^(qs-1) - 1) M
  int dstride;    // doubles between consecutive blocks of the step-0 input vectors d0 / d0b (0: M, contiguous)
  int ostride;    // same for the boundary rows a forward stage writes (partR / partL); 2 M = interleaved per chunk
  const double* lu_last;
  const int32_t* perm_last;
#ifdef AGGMG_CR_TRACE
  unsigned long long* trace;  // [3 kinds][kCrTraceWgs][16] constant-clock stamps (tools/cr_trace.py; never in the product build)
  int trace_kind;
#endif
};

#ifdef AGGMG_CR_TRACE
constexpr int kCrTraceWgs = 4096;
// stamps go to LDS and leave for global memory when the launch ends: a global store per stamp would sit in front of the
// next barrier's vmcnt(0) and show up as a memory round trip in every step
__device__ __forceinline__ unsigned long long* cr_trace_slots() {
  __shared__ unsigned long long t[16];
  return t;
}
#define CR_STAMP(A, slot)                                                        \
  do {                                                                           \
    if (threadIdx.x == 0) cr_trace_slots()[slot] = wall_clock64();               \
  } while (0)
#define CR_STAMP_WAIT(A, slot)                                                                              \
  do {                                                                                                      \
    if (threadIdx.x == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                      \
    CR_STAMP(A, slot);                                                                                      \
  } while (0)
#define CR_TRACE_BEGIN()                                                         \
  do {                                                                           \
    if (threadIdx.x < 16) cr_trace_slots()[threadIdx.x] = 0;                     \
  } while (0)
#define CR_TRACE_FLUSH(A)                                                                                             \
  do {                                                                                                                \
    if (threadIdx.x < 16 && (A).trace && blockIdx.x < kCrTraceWgs)                                                    \
      (A).trace[(((A).trace_kind * kCrTraceWgs) + blockIdx.x) * 16 + threadIdx.x] = cr_trace_slots()[threadIdx.x];    \
  } while (0)
#else
#define CR_STAMP(A, slot) ((void)0)
#define CR_STAMP_WAIT(A, slot) ((void)0)
#define CR_TRACE_BEGIN() ((void)0)
#define CR_TRACE_FLUSH(A) ((void)0)
#endif

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
  // (the factors carry the pivots of U as reciprocals: a fp64 division is a dependent chain of a dozen instructions, and
  // the small steps of a launch are one wave issuing such chains -- 14 divisions per sub-chunk and pass at block size 2)
  if constexpr (M == 1) {
    y[0] = r[0] * lu[0];
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
      y[i] = s * f[i * M + i];
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
    // the odd rows' right-hand sides, each used by both even neighbours; zero for rows that do not exist
    double y[NB1][M];
#pragma unroll
    for (int jj = 0; jj < NB1; ++jj) {
      const bool ok = tlo + 2 * jj + 1 <= thi;
#pragma unroll
      for (int e = 0; e < M; ++e) y[jj][e] = ok ? d[(2 * jj + 1) * M + e] : 0.0;
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

// ---- block sizes 1 and 2: every factor of a sub-chunk fetched in ONE batch --------------------------------------
// Stamps written from inside the launches (tools/cr_trace.py) show a wave spending 13.6 us between "input vector
// loaded" and "three sub-levels done" at block size 2: the compiler, minding its register budget, issues the ~60
// factor loads of a sub-chunk in about ten dependent batches, each a full memory round trip (cold in a cycle), and
// fetches the level descriptors (kernel arguments: scalar loads, a round trip each) as it goes.  Up to block size 2
// the factors of all three sub-levels fit the register file (forward: the multipliers of 10 even rows, 80 doubles at block
// size 2; backward: 7 odd rows, 84 doubles + 14 ints), so they are loaded up front, behind a scheduling barrier, and the
// arithmetic starts when they are all on their way: two round trips per step (descriptors, then factors and input).
template <int QS, int I>
struct CrEvenOff {
  static constexpr int rows = CrEvenOff<QS, I - 1>::rows + (1 << (QS - I)) + 1;
};
template <int QS>
struct CrEvenOff<QS, 0> {
  static constexpr int rows = 0;
};
template <int QS, int I>
struct CrOddOff {
  static constexpr int rows = CrOddOff<QS, I - 1>::rows + (1 << (QS - I));
};
template <int QS>
struct CrOddOff<QS, 0> {
  static constexpr int rows = 0;
};

template <int M, int QS>
struct CrSubFactors {
  double fe[CrEvenOff<QS, QS>::rows][2 * M * M];
};

template <int M, int QS, int I>
__device__ __forceinline__ void cr_pre_load(const CrLevel* lv, int64_t b, CrSubFactors<M, QS>& F) {
  if constexpr (I < QS) {
    const CrLevel& L = lv[I];
    constexpr int NB1 = 1 << (QS - I - 1);
    const int64_t tlo1 = b << (QS - I - 1);
#pragma unroll
    for (int jj = 0; jj <= NB1; ++jj) {
      int64_t j = tlo1 + jj;
      if (j > L.n_even - 1) j = L.n_even - 1;
      cr_load_pair<M>(L.fe + j * (2 * M * M), F.fe[CrEvenOff<QS, I>::rows + jj]);
    }
    cr_pre_load<M, QS, I + 1>(lv, b, F);
  }
}

// cr_loc_fwd on fetched factors (same operations in the same order: bit for bit the same result)
template <int M, int QS, int I>
__device__ __forceinline__ void cr_loc_fwd_pre(const CrLevel* lv, const CrSubFactors<M, QS>& F, int64_t b, double* v) {
  if constexpr (I < QS) {
    const CrLevel& L = lv[I];
    constexpr int NB1 = 1 << (QS - I - 1);
    int64_t thi = (b + 1) << (QS - I);
    if (thi > L.n - 1) thi = L.n - 1;
    const int64_t tlo = b << (QS - I);
    double* d = v + CrOff<QS, I>::blocks * M;
    double* dn = v + CrOff<QS, I + 1>::blocks * M;
    double y[NB1][M];
#pragma unroll
    for (int jj = 0; jj < NB1; ++jj) {
      const bool ok = tlo + 2 * jj + 1 <= thi;
#pragma unroll
      for (int e = 0; e < M; ++e) y[jj][e] = ok ? d[(2 * jj + 1) * M + e] : 0.0;
    }
#pragma unroll
    for (int jj = 0; jj <= NB1; ++jj) {
      const double* ac = F.fe[CrEvenOff<QS, I>::rows + jj];
      double acc[M];
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
    cr_loc_fwd_pre<M, QS, I + 1>(lv, F, b, v);
  }
}

// inner sub-levels' odd rows of a sub-chunk: sub-level I (1 <= I < QS) holds 2^(QS-I-1) of them
template <int QS, int I>
struct CrMidOff {
  static constexpr int blocks = CrMidOff<QS, I - 1>::blocks + (1 << (QS - I));
};
template <int QS>
struct CrMidOff<QS, 1> {
  static constexpr int blocks = 0;
};
template <int QS>
struct CrMidOff<QS, 0> {
  static constexpr int blocks = 0;
};
template <int QS>
constexpr int cr_mid_blocks() {
  return (1 << (QS - 1)) - 1;
}

template <int M, int QS, int I, bool STORE>
__device__ __forceinline__ void cr_mid_move(double* mid, double* v) {
  if constexpr (I < QS) {
    constexpr int NB1 = 1 << (QS - I - 1);
#pragma unroll
    for (int jj = 0; jj < NB1; ++jj)
#pragma unroll
      for (int e = 0; e < M; ++e) {
        double& r = v[(CrOff<QS, I>::blocks + 2 * jj + 1) * M + e];
        double& g = mid[(CrMidOff<QS, I>::blocks + jj) * M + e];
        if constexpr (STORE) g = r; else r = g;
      }
    cr_mid_move<M, QS, I + 1, STORE>(mid, v);
  }
}

// what the back substitution of a sub-chunk reads, fetched in one batch (block sizes 1 and 2, as above)
template <int M, int QS>
struct CrSubFactorsB {
  double fo[CrOddOff<QS, QS>::rows][2 * M * M];
  double lu[CrOddOff<QS, QS>::rows][M * M];
  int32_t perm[CrOddOff<QS, QS>::rows][M];
};

template <int M, int QS, int I>
__device__ __forceinline__ void cr_pre_load_b(const CrLevel* lv, int64_t b, CrSubFactorsB<M, QS>& F) {
  if constexpr (I < QS) {
    const CrLevel& L = lv[I];
    constexpr int NB1 = 1 << (QS - I - 1);
    const int64_t tlo1 = b << (QS - I - 1);
#pragma unroll
    for (int jj = 0; jj < NB1; ++jj) {
      int64_t idx = tlo1 + jj;
      if (idx > L.n_odd - 1) idx = L.n_odd - 1;
      const double* lu = L.lu + idx * (M * M);
      const int32_t* pm = L.perm + idx * M;
      constexpr int o = CrOddOff<QS, I>::rows;
      cr_load_pair<M>(L.fo + idx * (2 * M * M), F.fo[o + jj]);
      if constexpr (M == 2) {
        const double2 t0 = reinterpret_cast<const double2*>(lu)[0], t1 = reinterpret_cast<const double2*>(lu)[1];
        const int2 pp = *reinterpret_cast<const int2*>(pm);
        F.lu[o + jj][0] = t0.x, F.lu[o + jj][1] = t0.y, F.lu[o + jj][2] = t1.x, F.lu[o + jj][3] = t1.y;
        F.perm[o + jj][0] = pp.x, F.perm[o + jj][1] = pp.y;
      } else {
#pragma unroll
        for (int k = 0; k < M * M; ++k) F.lu[o + jj][k] = lu[k];
#pragma unroll
        for (int k = 0; k < M; ++k) F.perm[o + jj][k] = pm[k];
      }
    }
    cr_pre_load_b<M, QS, I + 1>(lv, b, F);
  }
}

// cr_loc_bwd on fetched factors
template <int M, int QS, int I>
__device__ __forceinline__ void cr_loc_bwd_pre(const CrLevel* lv, const CrSubFactorsB<M, QS>& F, int64_t b, double* v) {
  if constexpr (I >= 0) {
    const CrLevel& L = lv[I];
    constexpr int NB1 = 1 << (QS - I - 1);
    const int64_t tlo = b << (QS - I);
    double* d = v + CrOff<QS, I>::blocks * M;
    const double* xn = v + CrOff<QS, I + 1>::blocks * M;
#pragma unroll
    for (int jj = 0; jj < NB1; ++jj) {
      const int64_t r = tlo + 2 * jj + 1;
      const double* ac = F.fo[CrOddOff<QS, I>::rows + jj];
      double rhs[M], x[M];
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
      cr_lu_solve_reg<M>(F.lu[CrOddOff<QS, I>::rows + jj], F.perm[CrOddOff<QS, I>::rows + jj], rhs, x);
#pragma unroll
      for (int e = 0; e < M; ++e) d[(2 * jj + 1) * M + e] = x[e];
    }
#pragma unroll
    for (int jj = 0; jj <= NB1; ++jj)
#pragma unroll
      for (int e = 0; e < M; ++e) d[(2 * jj) * M + e] = xn[jj * M + e];
    cr_loc_bwd_pre<M, QS, I - 1>(lv, F, b, v);
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

template <int M, int QS>
__device__ __forceinline__ void cr_step_forward(const CrStageArgs& A, int s, int64_t c, bool wg_shared,
                                                const double* __restrict__ d0, const double* __restrict__ d0b,
                                                double* sh) {
  const CrStepGeom g = cr_step_geom<QS>(A, s, c, wg_shared);
  const int cnt_out = ((1 << (A.q - g.a1)) + 1) * M;
  double* Rout = sh + A.lds_off[s + 1];
  double* Lout = Rout + cnt_out;
  for (int i = threadIdx.x; i < g.nb; i += blockDim.x) {
    const int64_t b = g.lo_out + i;
    double v[CrOff<QS, QS + 1>::blocks * M];
    int64_t thi;
    bool tshared;
    if constexpr (M <= 2) {
      CrSubFactors<M, QS> F;
      cr_pre_load<M, QS, 0>(&A.lv[g.a], b, F);
      __builtin_amdgcn_sched_barrier(0);
      cr_loc_load<M, QS>(A, s, g, b, d0, d0b, sh, v, thi, tshared, wg_shared && s > 0 && i == g.nb - 1);
#ifdef AGGMG_CR_TRACE
      if (s == 0 && i == (int)threadIdx.x) CR_STAMP_WAIT(A, 2);
#endif
      cr_loc_fwd_pre<M, QS, 0>(&A.lv[g.a], F, b, v);
    } else {
      cr_loc_load<M, QS>(A, s, g, b, d0, d0b, sh, v, thi, tshared, wg_shared && s > 0 && i == g.nb - 1);
      cr_loc_fwd<M, QS, 0>(&A.lv[g.a], b, v);
    }
    if constexpr (QS >= 2) cr_mid_move<M, QS, 1, true>(A.mid + A.mid_off[s] + b * (cr_mid_blocks<QS>() * M), v);
    const double* top = v + CrOff<QS, QS>::blocks * M;
#ifdef AGGMG_CR_TRACE
    if (s == 0 && i == (int)threadIdx.x) {
      double t0 = top[0];
      asm volatile("" : "+v"(t0));
      CR_STAMP_WAIT(A, 3);
    }
#endif
#pragma unroll
    for (int e = 0; e < M; ++e) Rout[i * M + e] = top[e];
    if (b + 1 <= g.n_out - 1) {
#pragma unroll
      for (int e = 0; e < M; ++e) Lout[(i + 1) * M + e] = top[M + e];
    }
  }
}

// xtop: x of the step's output level, indexed by block - xtop_lo (LDS, or the global boundary
// solution); xout: x of the step's input level (LDS for s >= 1, the caller's vector for s == 0)
template <int M, int QS>
__device__ __forceinline__ void cr_step_backward(const CrStageArgs& A, int s, int64_t c, bool wg_shared,
                                                 const double* __restrict__ d0, const double* __restrict__ d0b,
                                                 const double* xtop, int64_t xtop_lo, double* xout, int64_t xout_lo,
                                                 double* sh) {
  constexpr int NB = 1 << QS;
  const CrStepGeom g = cr_step_geom<QS>(A, s, c, wg_shared);
  for (int i = threadIdx.x; i < g.nb; i += blockDim.x) {
    const int64_t b = g.lo_out + i;
    double v[CrOff<QS, QS + 1>::blocks * M];
    int64_t thi;
    bool tshared;
    // sub-level 0 from the step's input, the inner sub-levels' odd rows as the forward step left them
    double* top = v + CrOff<QS, QS>::blocks * M;
    const bool has_right = b + 1 <= g.n_out - 1;
    const int64_t br = has_right ? b + 1 : b;
    if constexpr (M <= 2) {
      CrSubFactorsB<M, QS> F;
      cr_pre_load_b<M, QS, 0>(&A.lv[g.a], b, F);
      __builtin_amdgcn_sched_barrier(0);
      cr_loc_load<M, QS>(A, s, g, b, d0, d0b, sh, v, thi, tshared, wg_shared && s > 0 && i == g.nb - 1);
      if constexpr (QS >= 2) cr_mid_move<M, QS, 1, false>(A.mid + A.mid_off[s] + b * (cr_mid_blocks<QS>() * M), v);
#pragma unroll
      for (int e = 0; e < M; ++e) {
        top[e] = xtop[(b - xtop_lo) * M + e];
        const double xr = xtop[(br - xtop_lo) * M + e];
        top[M + e] = has_right ? xr : 0.0;
      }
      cr_loc_bwd_pre<M, QS, QS - 1>(&A.lv[g.a], F, b, v);
    } else {
      cr_loc_load<M, QS>(A, s, g, b, d0, d0b, sh, v, thi, tshared, wg_shared && s > 0 && i == g.nb - 1);
      if constexpr (QS >= 2) cr_mid_move<M, QS, 1, false>(A.mid + A.mid_off[s] + b * (cr_mid_blocks<QS>() * M), v);
#pragma unroll
      for (int e = 0; e < M; ++e) {
        top[e] = xtop[(b - xtop_lo) * M + e];
        const double xr = xtop[(br - xtop_lo) * M + e];
        top[M + e] = has_right ? xr : 0.0;
      }
      cr_loc_bwd<M, QS, QS - 1>(&A.lv[g.a], b, v);
    }
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
}

template <int M>
__device__ __forceinline__ void cr_forward_steps(const CrStageArgs& A, int64_t c, bool wg_shared,
                                                 const double* __restrict__ d0, const double* __restrict__ d0b,
                                                 double* sh) {
  constexpr int Q = CrRadix<M>::Q;
  for (int s = 0; s < A.nsteps; ++s) {
    const int qs = A.step_a[s + 1] - A.step_a[s];
    if (qs == 1) {
      cr_step_forward<M, 1>(A, s, c, wg_shared, d0, d0b, sh);
    } else if (qs == 2) {
      cr_step_forward<M, 2>(A, s, c, wg_shared, d0, d0b, sh);
    } else {
      if constexpr (Q >= 3) cr_step_forward<M, 3>(A, s, c, wg_shared, d0, d0b, sh);
    }
    __syncthreads();
    CR_STAMP(A, 4 + s);
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
    if (qs == 1) {
      cr_step_backward<M, 1>(A, s, c, wg_shared, d0, d0b, xtop, xtop_lo, xout, xout_lo, sh);
    } else if (qs == 2) {
      cr_step_backward<M, 2>(A, s, c, wg_shared, d0, d0b, xtop, xtop_lo, xout, xout_lo, sh);
    } else {
      if constexpr (Q >= 3) cr_step_backward<M, 3>(A, s, c, wg_shared, d0, d0b, xtop, xtop_lo, xout, xout_lo, sh);
    }
    __syncthreads();
    CR_STAMP(A, 8 + s);
  }
}

// the calling workgroup solves the whole tail system (T.tail) for d0 (+ d0b) into x0
template <int M>
__device__ __forceinline__ void cr_tail_body(const CrStageArgs& T, const double* __restrict__ d0,
                                             const double* __restrict__ d0b, double* __restrict__ x0, double* sh) {
  CR_TRACE_BEGIN();
  CR_STAMP(T, 0);
  for (int t = threadIdx.x; t < T.lds_total; t += blockDim.x) sh[t] = 0.0;
  __syncthreads();
  CR_STAMP(T, 1);
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
  CR_STAMP(T, 12);
  cr_backward_steps<M>(T, 0, false, d0, d0b, nullptr, x0, sh);
  CR_STAMP_WAIT(T, 15);
  CR_TRACE_FLUSH(T);
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
  CR_TRACE_BEGIN();
  CR_STAMP(A, 0);
  for (int t = threadIdx.x; t < A.lds_total; t += blockDim.x) sh[t] = 0.0;
  __syncthreads();
  CR_STAMP(A, 1);
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
  CR_STAMP_WAIT(A, 15);
  CR_TRACE_FLUSH(A);
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
  CR_TRACE_BEGIN();
  CR_STAMP(A, 0);
  for (int t = threadIdx.x; t < A.lds_total; t += blockDim.x) sh[t] = 0.0;
  __syncthreads();
  CR_STAMP(A, 1);
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
      CR_STAMP(A, 2);
    } else {
      cr_forward_steps<M>(A, c, wg_shared, d0, d0b, sh);
    }
  }
  cr_backward_steps<M>(A, c, wg_shared, d0, d0b, xq, x0, sh);
  CR_STAMP_WAIT(A, 15);
  CR_TRACE_FLUSH(A);
}

// ---- the tail by PARALLEL cyclic reduction ---------------------------------------------------------------------------
// The boundary system the last stage leaves (a few hundred blocks) is a chain of log2(n) dependent levels either way;
// the register-blocked cyclic reduction above takes them three at a time with 32, 4, 1 threads working and pays 1.6 - 3.5 us
// per step down and again per step up (tools/cr_trace.py).  Parallel cyclic reduction keeps every row through all levels:
// at level k row i drops its couplings to rows i -+ 2^k,
//     d_i <- d_i - alpha_i^k d_{i-2^k} - gamma_i^k d_{i+2^k},
// with the multipliers alpha = a b^-1, gamma = c b^-1 of every (level, row) formed at set-up (pcr_reduce_kernel), after
// ceil(log2 n) levels every row stands alone, x_i = b_i \ d_i: one thread per row, two small matrix-vector products and a
// barrier per level, no way back up.  Up to 512 rows; 513 .. 1024 rows take one ordinary cyclic-reduction level first (PRE).
constexpr int kPcrMaxLevels = 9;  // up to 512 rows in the parallel part, one thread each
struct PcrArgs {
  const double* mult;   // [L][n][2][M][M]
  const double* lu;     // [n][M][M]  diagonal blocks after the last level: pivoted LU, reciprocal pivots
  const int32_t* perm;  // [n][M]
  int n, L;
  int dstride;          // doubles between consecutive blocks of d0 / d0b (0: M)
  // PRE: one level of ordinary cyclic reduction around the parallel part (513 .. 1024 rows): thread t takes even row 2t through
  // the parallel levels and solves odd row 2t + 1 from its two even neighbours afterwards -- the level's own factors
  CrLevel lv0;
  int n_full;
#ifdef AGGMG_CR_TRACE
  unsigned long long* trace;
#endif
};

// Every multiplier a thread needs is fetched in ONE batch before the first level (512 threads: two waves per SIMD, 256
// registers each -- nine levels of two M x M blocks at block size 2 are 144 of them).
template <int M, bool PRE>
__global__ __launch_bounds__(512) void cr_pcr_tail_kernel(PcrArgs P, const double* __restrict__ d0,
                                                           const double* __restrict__ d0b, double* __restrict__ x0) {
  extern __shared__ double sh[];
  constexpr int W = 2 * M * M, D = kPcrMaxLevels;
  const int i = threadIdx.x, n = P.n;
  const bool act = i < n;
  const int ic = act ? i : n - 1;
#ifdef AGGMG_CR_TRACE
  unsigned long long t_in = 0;
  if (i == 0) t_in = wall_clock64();
#endif
  double f[D][W];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const int lv = j < P.L ? j : 0;
    const double* src = P.mult + ((int64_t)lv * n + ic) * W;
#pragma unroll
    for (int k = 0; k < W / 2; ++k) {
      const double2 t = reinterpret_cast<const double2*>(src)[k];
      f[j][2 * k] = t.x, f[j][2 * k + 1] = t.y;
    }
  }
  double luf[M * M];
  int32_t pf[M];
#pragma unroll
  for (int k = 0; k < M * M; ++k) luf[k] = P.lu[(int64_t)ic * (M * M) + k];
#pragma unroll
  for (int k = 0; k < M; ++k) pf[k] = P.perm[(int64_t)ic * M + k];
  const int ds = P.dstride ? P.dstride : M;
  double v[M];
  // PRE: the level's factors of even row 2 i (forward multipliers) and odd row 2 i + 1 (couplings, factored block), and the
  // right-hand sides of rows 2 i - 1, 2 i, 2 i + 1
  double fe0[PRE ? W : 1], fo0[PRE ? W : 1], lu0[PRE ? M * M : 1], dodd[M];
  int32_t p0[M];
  bool has_odd = false;
  if constexpr (PRE) {
    const int nf = P.n_full;
    const int64_t io = 2 * ic + 1 < nf ? ic : (P.lv0.n_odd > 0 ? P.lv0.n_odd - 1 : 0);  // odd row 2 ic + 1 (clamped)
    has_odd = act && 2 * i + 1 < nf;
    cr_load_pair<M>(P.lv0.fe + (int64_t)ic * W, fe0);
    cr_load_pair<M>(P.lv0.fo + io * W, fo0);
#pragma unroll
    for (int k = 0; k < M * M; ++k) lu0[k] = P.lv0.lu[io * (M * M) + k];
#pragma unroll
    for (int k = 0; k < M; ++k) p0[k] = P.lv0.perm[io * M + k];
    const int64_t rl = 2 * ic - 1 >= 0 ? 2 * ic - 1 : 0, rr = 2 * ic + 1 < nf ? 2 * ic + 1 : nf - 1;
    double dl[M], dr[M];
#pragma unroll
    for (int e = 0; e < M; ++e) {
      v[e] = d0[(int64_t)(2 * ic) * ds + e];
      dl[e] = d0[rl * ds + e];
      dr[e] = d0[rr * ds + e];
    }
    if (d0b) {
#pragma unroll
      for (int e = 0; e < M; ++e) {
        v[e] += d0b[(int64_t)(2 * ic) * ds + e];
        dl[e] += d0b[rl * ds + e];
        dr[e] += d0b[rr * ds + e];
      }
    }
#pragma unroll
    for (int e = 0; e < M; ++e) dodd[e] = dr[e];
    // (a missing neighbour: its multiplier is zero)
#pragma unroll
    for (int r = 0; r < M; ++r) {
      double t = v[r];
#pragma unroll
      for (int q = 0; q < M; ++q) t -= fe0[r * M + q] * dl[q];
#pragma unroll
      for (int q = 0; q < M; ++q) t -= fe0[M * M + r * M + q] * dr[q];
      v[r] = t;
    }
  } else {
#pragma unroll
    for (int e = 0; e < M; ++e) v[e] = d0[(int64_t)ic * ds + e];
    if (d0b) {
#pragma unroll
      for (int e = 0; e < M; ++e) v[e] += d0b[(int64_t)ic * ds + e];
    }
  }
  double* cur = sh;
  double* nxt = sh + (int64_t)(n + 1) * M;
  if (act) {
#pragma unroll
    for (int e = 0; e < M; ++e) cur[i * M + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPcrMaxLevels; ++k) {
    if (k < P.L) {
      const int s = 1 << k;
      const int il = ic - s >= 0 ? ic - s : 0, ir = ic + s < n ? ic + s : n - 1;  // (missing neighbours: zero multipliers)
      double dl[M], dr[M];
#pragma unroll
      for (int e = 0; e < M; ++e) dl[e] = cur[il * M + e], dr[e] = cur[ir * M + e];
#pragma unroll
      for (int r = 0; r < M; ++r) {
        double t = v[r];
#pragma unroll
        for (int q = 0; q < M; ++q) t -= f[k][r * M + q] * dl[q];
#pragma unroll
        for (int q = 0; q < M; ++q) t -= f[k][M * M + r * M + q] * dr[q];
        v[r] = t;
      }
      if (k + 1 < P.L) {
        if (act) {
#pragma unroll
          for (int e = 0; e < M; ++e) nxt[i * M + e] = v[e];
        }
        __syncthreads();
        double* t = cur;
        cur = nxt;
        nxt = t;
      }
    }
  }
  double y[M];
  cr_lu_solve_reg<M>(luf, pf, v, y);
  if constexpr (PRE) {
    // x of the even rows to LDS (one more slot: the right neighbour of the last odd row may not exist), odd rows from them
    __syncthreads();
    if (act) {
#pragma unroll
      for (int e = 0; e < M; ++e) nxt[i * M + e] = y[e];
    }
    if (i == 0) {
#pragma unroll
      for (int e = 0; e < M; ++e) nxt[n * M + e] = 0.0;
    }
    __syncthreads();
    if (act) {
#pragma unroll
      for (int e = 0; e < M; ++e) x0[(int64_t)(2 * i) * M + e] = y[e];
    }
    if (has_odd) {
      double rhs[M], xo[M];
#pragma unroll
      for (int r = 0; r < M; ++r) {
        double t = dodd[r];
#pragma unroll
        for (int q = 0; q < M; ++q) t -= fo0[r * M + q] * y[q];
#pragma unroll
        for (int q = 0; q < M; ++q) t -= fo0[M * M + r * M + q] * nxt[(i + 1) * M + q];
        rhs[r] = t;
      }
      cr_lu_solve_reg<M>(lu0, p0, rhs, xo);
#pragma unroll
      for (int e = 0; e < M; ++e) x0[(int64_t)(2 * i + 1) * M + e] = xo[e];
    }
  } else {
    if (act) {
#pragma unroll
      for (int e = 0; e < M; ++e) x0[(int64_t)i * M + e] = y[e];
    }
  }
#ifdef AGGMG_CR_TRACE
  if (i == 0 && P.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P.trace[(1 * kCrTraceWgs + 0) * 16 + 0] = t_in;
    P.trace[(1 * kCrTraceWgs + 0) * 16 + 15] = wall_clock64();
  }
#endif
}

}  // namespace aggmg
