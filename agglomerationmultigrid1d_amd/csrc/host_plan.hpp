// Host-only planning arithmetic of libaggmg_hip.so -- no HIP types, no device calls: the pieces of the launch logic
// that decide index ranges, LDS layouts, tile subsets and copy slices.  Kept apart so that a host-compiled test
// (tests/host/test_host_plan.cpp) can drive them under AddressSanitizer / UBSan without a device (SURVEY.md
// section 5: sanitizers on the CPU build only); the .hip files use exactly these functions.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace aggmg {

constexpr int kCrTailRows = 4096;  // scalar rows (blocks * m) the single-workgroup tail takes
constexpr int kCrMaxLevels = 40;
constexpr int kCrMaxStageLevels = 12;
constexpr int kCrMaxSteps = 8;

// ---- coarsest solve (src/solvers.jl:39): one launch of the cyclic reduction -----------------------------------
// levels [l0, l0 + q) of the reduction in steps of up to three thread-local levels (cr_kernels.hpp): step split,
// LDS layout, strides of the per-chunk stack and of the stored inner right-hand sides
struct CrStagePlan {
  int l0 = 0, q = 0;
  int64_t n_in = 0, n_out = 0;  // blocks before / after
  int nsteps = 0;
  int step_a[kCrMaxSteps + 1] = {0};
  int lds_off[kCrMaxSteps + 1] = {0}, lds_xoff[kCrMaxSteps + 1] = {0};
  int lds_total = 0;            // doubles
  int stack_stride = 0;
  int64_t mid_off[kCrMaxSteps + 1] = {0};
  int64_t mid_total = 0;        // doubles
};

// step split and LDS layout of a stage of q levels with block size m
inline void cr_plan_steps(CrStagePlan* S, int m) {
  const int Q = m <= 4 ? 3 : 2;
  S->nsteps = 0;
  S->step_a[0] = 0;
  for (int a = 0; a < S->q;) {
    const int qs = std::min(Q, S->q - a);
    a += qs;
    S->step_a[++S->nsteps] = a;
  }
  int o = 0;
  for (int s = 1; s <= S->nsteps; ++s) {
    const int cnt = ((1 << (S->q - S->step_a[s])) + 1) * m;
    S->lds_off[s] = o;
    o += 2 * cnt;
  }
  for (int s = 1; s <= S->nsteps; ++s) {
    const int cnt = ((1 << (S->q - S->step_a[s])) + 1) * m;
    S->lds_xoff[s] = o;
    o += cnt;
  }
  S->lds_total = o;
  S->stack_stride = 0;
  for (int s = 1; s < S->nsteps; ++s) S->stack_stride += ((1 << (S->q - S->step_a[s])) + 1) * m;
  // sub-chunk b of step s (a block index of the step's output level: at most (n_in >> a1) + 1 of them) keeps
  // 2^(qs-1) - 1 blocks; every region 16-byte aligned
  S->mid_total = 0;
  for (int s = 0; s < S->nsteps; ++s) {
    const int qs = S->step_a[s + 1] - S->step_a[s];
    S->mid_off[s] = S->mid_total;
    const int64_t cnt = ((S->n_in >> S->step_a[s + 1]) + 2) * (((int64_t)1 << (qs - 1)) - 1) * m;
    S->mid_total += (cnt + 1) & ~(int64_t)1;
  }
}

// LDS of a stage of q levels: two right-hand-side vectors and a solution per block of every step's output level, the
// first step's (2^(q - Q) + 1 blocks) dominating.  A stage must fit the LDS of a compute unit (gfx950: 160 KB, one
// workgroup per CU then): chunks of 2^12 blocks do up to block size 4 (55 KB); block sizes 5 - 8 would ask for
// 164 ... 262 KB -- a launch failure, on systems of 2^20 blocks and more -- and take chunks of 2^11.
constexpr size_t kCrLdsBudgetBytes = 160 * 1024;
inline size_t cr_stage_lds_bytes(int q, int m) {
  CrStagePlan S;
  S.q = q;
  cr_plan_steps(&S, m);
  return (size_t)S.lds_total * sizeof(double);
}
inline int cr_max_stage_levels(int m) {
  int q = kCrMaxStageLevels;
  while (q > 1 && cr_stage_lds_bytes(q, m) > kCrLdsBudgetBytes) --q;
  return q;
}

// Which launches a solve takes: chunk stages (one workgroup per 2^q-block chunk) until what is left fits the
// single-workgroup tail.  level_n[l] = blocks of reduction level l (level_n.size() = nl reducing levels; beyond
// them one block).  fill_max / min_workgroups: large chunks as long as a few hundred workgroups remain.
// -> false when the system cannot be planned (too many levels left for the tail).
struct CrSolvePlan {
  std::vector<CrStagePlan> stages;
  CrStagePlan tail;
};
inline bool cr_plan_solve(const std::vector<int64_t>& level_n, int m, int tail_rows, int max_q, int fill_max,
                          int min_workgroups, CrSolvePlan* out) {
  const int nl = (int)level_n.size();
  auto ln = [&](int l) -> int64_t { return l < nl ? level_n[l] : 1; };
  out->stages.clear();
  max_q = std::min(max_q, cr_max_stage_levels(m));
  int l0 = 0;
  while (l0 < nl && ln(l0) * m > tail_rows) {
    int need = 0;
    while (l0 + need < nl && ln(l0 + need) * m > tail_rows) ++need;
    int fill = 0;
    while (fill < fill_max && (ln(l0) >> (fill + 1)) >= min_workgroups) ++fill;
    CrStagePlan S;
    S.l0 = l0;
    S.q = std::min({max_q, std::max(need, fill), nl - l0});
    if (S.q < 1) break;
    S.n_in = ln(l0);
    S.n_out = ln(l0 + S.q);
    cr_plan_steps(&S, m);
    out->stages.push_back(S);
    l0 += S.q;
  }
  if (nl - l0 > kCrMaxStageLevels || ln(l0) * m > tail_rows) return false;
  out->tail = CrStagePlan();
  out->tail.l0 = l0;
  out->tail.q = nl - l0;
  out->tail.n_in = ln(l0);
  out->tail.n_out = 1;
  cr_plan_steps(&out->tail, m);
  return true;
}

// ---- generic CSR kernels: row blocks -------------------------------------------------------------------------------
// csr_stream_kernel: runs of consecutive rows holding at most max_nnz entries and at most max_rows rows; a row longer
// than max_nnz stands alone (the kernel reduces it across the workgroup).  -> block boundaries, blocks + 1 entries
inline std::vector<int32_t> stream_row_blocks(const int32_t* rowptr, int64_t nrows, int max_nnz, int max_rows) {
  std::vector<int32_t> blk;
  blk.push_back(0);
  int64_t r = 0;
  while (r < nrows) {
    int64_t e = r + 1;  // a block always takes at least one row
    const int64_t b0 = rowptr[r];
    while (e < nrows && e - r < max_rows && rowptr[e + 1] - b0 <= max_nnz) ++e;
    if (rowptr[r + 1] - b0 > max_nnz) e = r + 1;
    blk.push_back((int32_t)e);
    r = e;
  }
  return blk;
}

// csr_band_kernel (entries within bw of the diagonal, up to `sweeps` point-Jacobi sweeps per launch): a block's rows plus
// (sweeps - 1) * bw halo rows on either side hold at most max_nnz entries, and its window of x -- the rows plus
// sweeps * bw on either side -- fits `window` doubles.  -> false when some row is too long for its halo to fit
inline bool band_row_blocks(const int32_t* rowptr, int64_t nrows, int bw, int sweeps, int max_nnz, int window, int max_rows,
                            std::vector<int32_t>* out) {
  const int64_t H = (int64_t)(sweeps - 1) * bw;
  out->clear();
  out->push_back(0);
  int64_t r = 0;
  while (r < nrows) {
    const int64_t lo = std::max<int64_t>(0, r - H);
    auto fits = [&](int64_t e) {
      const int64_t hi = std::min(nrows, e + H);
      return rowptr[hi] - rowptr[lo] <= max_nnz && (e - r) + 2 * (int64_t)sweeps * bw <= window;
    };
    int64_t e = r + 1;
    if (!fits(e)) return false;
    while (e < nrows && e - r < max_rows && fits(e + 1)) ++e;
    out->push_back((int32_t)e);
    r = e;
  }
  return true;
}

// ---- fused level launches: which tiles a launch runs ------------------------------------------------------------
// A level of ne elements in tiles of `owned` elements.  mode 0: all tiles; 1: the tiles holding [0, head) and
// [tail, ne) (element-partitioned runs produce the interface elements first); 2: the ones in between.
// Workgroup b runs tile b + (b >= split ? skip : 0).
struct TileSubset {
  int64_t ntiles = 0;  // workgroups of the launch
  int split = 0;
  int64_t skip = 0;
};
inline TileSubset fused_tile_subset(int64_t ne, int owned, int mode, int64_t head_in, int64_t tail_in) {
  TileSubset t;
  t.ntiles = (ne + owned - 1) / owned;
  if (mode == 0) return t;
  const int64_t head = std::min(std::max<int64_t>(head_in, 0), ne);
  const int64_t tail = std::min(std::max(tail_in, head), ne);
  const int64_t all = t.ntiles;
  const int64_t tA = std::min(all, (head + owned - 1) / owned);   // tiles [0, tA) hold [0, head)
  const int64_t tB = std::min(all - tA, all - tail / owned);      // the last tB tiles hold [tail, ne)
  if (mode == 1) {
    t.split = (int)tA;
    t.skip = all - tA - tB;
    t.ntiles = tA + tB;
  } else {
    t.skip = tA;
    t.ntiles = all - tA - tB;
  }
  return t;
}

// ---- host-pointer entry (aggmg_vcycle): lane t's byte range of a copy split over `lanes` worker threads ----------
inline void stage_lane_range(size_t bytes, int lanes, int t, size_t* lo, size_t* hi) {
  const size_t per = ((bytes / (size_t)lanes) + 4095) & ~(size_t)4095;
  *lo = std::min(bytes, (size_t)t * per);
  *hi = t == lanes - 1 ? bytes : std::min(bytes, (size_t)(t + 1) * per);
}

// ---- element-partitioned coarsest solve: chunked or gathered? ---------------------------------------------------
// Decided from GLOBAL quantities only (every rank must take the same route: mismatched collective counts hang):
// the replicated operator's plan (q = log2 chunk, or <= 0: none; its block size and block count must be the
// level's) and an even division of the level into whole chunks per rank.
//  -> 0 gather-and-replicate, 1 chunked, -1 chunked applies but THIS rank's range breaks the pattern (refuse)
inline int dist_chunk_route(int q, int plan_m, int64_t plan_blocks, int m, int64_t ne, int world, int rank, int64_t own_lo,
                            int64_t own_hi) {
  if (!(q > 0 && plan_m == m && plan_blocks == ne) || world < 1 || ne % world != 0) return 0;
  const int64_t per_rank = ne / world, chunk = (int64_t)1 << q;
  if (per_rank % chunk != 0 || per_rank < chunk) return 0;
  if (own_hi - own_lo != per_rank || own_lo != (int64_t)rank * per_rank) return -1;
  return 1;
}

}  // namespace aggmg
