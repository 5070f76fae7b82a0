// Host-only unit test of the library's planning arithmetic (agglomerationmultigrid1d_amd/csrc/host_plan.hpp), built with
// g++ -fsanitize=address,undefined and run by tests/test_sanitizers_cpu.py -- no device, no HIP.  Every check is a
// property the kernels rely on: LDS regions that do not overlap and fit the CU, tile subsets that cover every tile
// exactly once, copy slices that partition the byte range, one chunk route for all ranks.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <utility>
#include <vector>

#include "../../agglomerationmultigrid1d_amd/csrc/host_plan.hpp"

using namespace aggmg;

static int failures = 0;
#define EXPECT(c)                                                          \
  do {                                                                     \
    if (!(c)) {                                                            \
      std::fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #c); \
      ++failures;                                                          \
    }                                                                      \
  } while (0)

static void check_stage(const CrStagePlan& S, int m) {
  EXPECT(S.nsteps >= (S.q > 0 ? 1 : 0) && S.nsteps <= kCrMaxSteps);
  EXPECT(S.step_a[0] == 0 && S.step_a[S.nsteps] == S.q);
  const int Q = m <= 4 ? 3 : 2;
  std::vector<char> used((size_t)S.lds_total, 0);
  auto claim = [&](int off, int cnt) {
    EXPECT(off >= 0 && off + cnt <= S.lds_total);
    for (int i = off; i < off + cnt && i < S.lds_total; ++i) {
      EXPECT(!used[(size_t)i]);
      used[(size_t)i] = 1;
    }
  };
  for (int s = 1; s <= S.nsteps; ++s) {
    EXPECT(S.step_a[s] - S.step_a[s - 1] >= 1 && S.step_a[s] - S.step_a[s - 1] <= Q);
    const int cnt = ((1 << (S.q - S.step_a[s])) + 1) * m;
    claim(S.lds_off[s], 2 * cnt);   // right-hand sides of the step's output level (two per block)
    claim(S.lds_xoff[s], cnt);      // its solution
  }
  for (char u : used) EXPECT(u);    // no holes either
  EXPECT((size_t)S.lds_total * sizeof(double) <= kCrLdsBudgetBytes);   // what a launch may ask for (the CU's 160 KB)
  int64_t prev_end = 0;
  for (int s = 0; s < S.nsteps; ++s) {
    EXPECT(S.mid_off[s] == prev_end && (S.mid_off[s] & 1) == 0);   // regions back to back, 16-byte aligned
    const int qs = S.step_a[s + 1] - S.step_a[s];
    const int64_t need = ((S.n_in >> S.step_a[s + 1]) + 1) * (((int64_t)1 << (qs - 1)) - 1) * m;
    const int64_t end = s + 1 < S.nsteps ? S.mid_off[s + 1] : S.mid_total;
    EXPECT(end - S.mid_off[s] >= need);
    prev_end = end;
  }
}

static void test_cr_plans() {
  for (int m = 1; m <= 8; ++m)
    for (int q = 1; q <= cr_max_stage_levels(m); ++q) {   // (the planner keeps q within the LDS budget)
      CrStagePlan S;
      S.q = q;
      S.n_in = ((int64_t)1 << q) * 37 + 5;
      S.n_out = S.n_in >> q;
      cr_plan_steps(&S, m);
      check_stage(S, m);
    }
  // whole solves: the systems of the benchmarked hierarchies and awkward sizes
  const int64_t sizes[] = {1, 2, 3, 255, 256, 257, 1000, 4096, 4097, 65536, 1 << 20, (1 << 20) + 17, 1 << 24, (1 << 24) - 1};
  for (int m : {1, 2, 3, 4, 8})
    for (int64_t n0 : sizes)
      for (int max_q : {1, 4, 10, 12})
        for (int tail_rows : {8, 64, 4096}) {
          std::vector<int64_t> ln;
          for (int64_t n = n0; n > 1; n = (n + 1) / 2) ln.push_back(n);   // n -> even rows ceil(n / 2)
          CrSolvePlan P;
          const bool ok = cr_plan_solve(ln, m, tail_rows, max_q, 12, 256, &P);
          if (!ok) continue;
          int l = 0;
          for (const CrStagePlan& S : P.stages) {
            EXPECT(S.l0 == l && S.q >= 1 && S.q <= max_q && S.q <= cr_max_stage_levels(m));
            EXPECT(S.n_in == (l < (int)ln.size() ? ln[(size_t)l] : 1));
            l += S.q;
            EXPECT(S.n_out == (l < (int)ln.size() ? ln[(size_t)l] : 1));
            check_stage(S, m);
          }
          EXPECT(P.tail.l0 == l && P.tail.q == (int)ln.size() - l && P.tail.q <= kCrMaxStageLevels);
          EXPECT(P.tail.n_in * m <= tail_rows || P.tail.n_in == 1);
          check_stage(P.tail, m);
        }
}

static void test_lds_cap() {
  EXPECT(cr_max_stage_levels(1) == 12 && cr_max_stage_levels(2) == 12 && cr_max_stage_levels(4) == 12);
  for (int m = 5; m <= 8; ++m) EXPECT(cr_max_stage_levels(m) == 11);   // 2^12-block chunks would ask for 164 ... 262 KB
  for (int m = 1; m <= 8; ++m) {
    const int q = cr_max_stage_levels(m);
    EXPECT(cr_stage_lds_bytes(q, m) <= kCrLdsBudgetBytes);
    EXPECT(q == kCrMaxStageLevels || cr_stage_lds_bytes(q + 1, m) > kCrLdsBudgetBytes);
  }
}

static void test_row_blocks() {
  // row-length patterns: uniform short rows (DG / CG operators), one long row among short ones, empty rows, growing rows
  unsigned seed = 12345u;
  auto rnd = [&](unsigned mod) { seed = seed * 1664525u + 1013904223u; return (seed >> 8) % mod; };
  for (int pattern = 0; pattern < 6; ++pattern)
    for (int64_t nrows : {(int64_t)1, (int64_t)7, (int64_t)1000, (int64_t)5000, (int64_t)70000}) {
      std::vector<int32_t> rp((size_t)nrows + 1, 0);
      for (int64_t i = 0; i < nrows; ++i) {
        int len = 0;
        switch (pattern) {
          case 0: len = 6; break;
          case 1: len = (i == nrows / 2) ? 10000 : 5; break;
          case 2: len = (int)rnd(3) == 0 ? 0 : 9; break;
          case 3: len = 1 + (int)(i % 40); break;
          case 4: len = (int)rnd(30); break;
          default: len = (i % 97 == 0) ? 5000 : 12; break;
        }
        rp[(size_t)i + 1] = rp[(size_t)i] + len;
      }
      const int max_nnz = 4096, max_rows = 1024;
      // (4096 / 1024: the r03 block shape; 1536 / 256: what csr_stream_kernel takes since r04 -- one row per thread)
      for (const auto& shape : {std::pair<int, int>{4096, 1024}, std::pair<int, int>{1536, 256}}) {
        const std::vector<int32_t> blk = stream_row_blocks(rp.data(), nrows, shape.first, shape.second);
        EXPECT(blk.front() == 0 && blk.back() == nrows);
        for (size_t k = 0; k + 1 < blk.size(); ++k) {
          const int r0 = blk[k], r1 = blk[k + 1];
          EXPECT(r1 > r0 && r1 - r0 <= shape.second);                                 // a partition into non-empty runs
          EXPECT(rp[(size_t)r1] - rp[(size_t)r0] <= shape.first || r1 == r0 + 1);     // that fit the LDS stage, or one long row
        }
      }
      // csr_band_kernel's tiles since r04 (setup.hip): S = 1 + 32 / bw sweeps at most 8, a block's rows plus (S - 1) bw halo
      // rows per side are at most 256 rows (window = 256 + 2 bw in band_row_blocks' terms) holding at most 2048 entries
      for (int bw : {1, 4, 7, 16, 32}) {
        const int S = std::max(2, std::min(8, 1 + 32 / bw));
        std::vector<int32_t> bb;
        if (!band_row_blocks(rp.data(), nrows, bw, S, 2048, 256 + 2 * bw, 256, &bb)) continue;
        EXPECT(bb.front() == 0 && bb.back() == nrows);
        const int64_t H = (int64_t)(S - 1) * bw;
        for (size_t k = 0; k + 1 < bb.size(); ++k) {
          const int64_t r0 = bb[k], r1 = bb[k + 1];
          EXPECT(r1 > r0);
          const int64_t lo = std::max<int64_t>(0, r0 - H), hi = std::min<int64_t>(nrows, r1 + H);
          EXPECT(hi - lo <= 256);                                    // one row of the tile per thread
          EXPECT(rp[(size_t)hi] - rp[(size_t)lo] <= 2048);           // 8 entries per thread in registers
        }
      }
      for (int bw : {1, 5, 32})
        for (int sweeps : {1, 4}) {
          const int window = 4 * 256 + 2 * 4 * 32;
          std::vector<int32_t> bb;
          const bool ok = band_row_blocks(rp.data(), nrows, bw, sweeps, max_nnz, window, max_rows, &bb);
          if (!ok) continue;
          EXPECT(bb.front() == 0 && bb.back() == nrows);
          const int64_t H = (int64_t)(sweeps - 1) * bw;
          for (size_t k = 0; k + 1 < bb.size(); ++k) {
            const int64_t r0 = bb[k], r1 = bb[k + 1];
            EXPECT(r1 > r0 && r1 - r0 <= max_rows);
            const int64_t lo = std::max<int64_t>(0, r0 - H), hi = std::min<int64_t>(nrows, r1 + H);
            EXPECT(rp[(size_t)hi] - rp[(size_t)lo] <= max_nnz);                     // block + halo rows fit the product stage
            EXPECT((r1 - r0) + 2 * (int64_t)sweeps * bw <= window);                 // and the window of x its LDS array
          }
        }
    }
}

static void test_tile_subsets() {
  for (int64_t ne : {1, 7, 100, 101, 1000, 4096, 100000})
    for (int owned : {1, 3, 64, 100, 122})
      for (int64_t head : {(int64_t)-5, (int64_t)0, (int64_t)1, (int64_t)80, ne / 2, ne, ne + 9})
        for (int64_t tail : {(int64_t)0, ne / 2, ne - 80, ne - 1, ne, ne + 3}) {
          const TileSubset all = fused_tile_subset(ne, owned, 0, head, tail);
          const TileSubset ends = fused_tile_subset(ne, owned, 1, head, tail);
          const TileSubset mid = fused_tile_subset(ne, owned, 2, head, tail);
          EXPECT(all.ntiles == (ne + owned - 1) / owned && all.split == 0 && all.skip == 0);
          std::multiset<int64_t> seen;
          for (const TileSubset* t : {&ends, &mid})
            for (int64_t b = 0; b < t->ntiles; ++b) seen.insert(b + (b >= t->split ? t->skip : 0));
          EXPECT((int64_t)seen.size() == all.ntiles);          // ends + middle = every tile ...
          int64_t want = 0;
          for (int64_t tile : seen) EXPECT(tile == want++);     // ... exactly once
          // the interface elements are in the "ends" launch
          const int64_t h = std::min(std::max<int64_t>(head, 0), ne), tl = std::min(std::max(tail, h), ne);
          std::set<int64_t> e;
          for (int64_t b = 0; b < ends.ntiles; ++b) e.insert(b + (b >= ends.split ? ends.skip : 0));
          for (int64_t x : {(int64_t)0, h - 1, tl, ne - 1})
            if (x >= 0 && x < ne && (x < h || x >= tl)) EXPECT(e.count(x / owned) == 1);
        }
}

static void test_lane_ranges() {
  for (size_t bytes : {(size_t)0, (size_t)1, (size_t)4095, (size_t)4096, (size_t)(16u << 20), (size_t)134217728, (size_t)134217729})
    for (int lanes = 1; lanes <= 8; ++lanes) {
      size_t at = 0;
      for (int t = 0; t < lanes; ++t) {
        size_t lo = 0, hi = 0;
        stage_lane_range(bytes, lanes, t, &lo, &hi);
        EXPECT(lo <= hi && hi <= bytes);
        if (hi > lo) {
          EXPECT(lo == at);               // slices back to back
          EXPECT(t == 0 || lo % 4096 == 0);   // on page boundaries
          at = hi;
        }
      }
      EXPECT(at == bytes);
    }
}

static void test_chunk_route() {
  // an even partition into whole chunks: every rank says "chunked"
  for (int world : {1, 2, 3, 4, 8})
    for (int q : {1, 4, 10}) {
      const int64_t per = ((int64_t)1 << q) * 3, ne = per * world;
      for (int r = 0; r < world; ++r) EXPECT(dist_chunk_route(q, 2, ne, 2, ne, world, r, r * per, (r + 1) * per) == 1);
      // same level, ranks owning 4, 3, 5 ... chunks: the ranks off the pattern are refused, none silently gathers
      if (world >= 2) {
        EXPECT(dist_chunk_route(q, 2, ne, 2, ne, world, 0, 0, per + ((int64_t)1 << q)) == -1);
        EXPECT(dist_chunk_route(q, 2, ne, 2, ne, world, 1, per + ((int64_t)1 << q), 2 * per) == -1);
      }
      // no plan, another block size, a level that does not divide: every rank gathers, whatever it owns
      for (int r = 0; r < world; ++r) {
        EXPECT(dist_chunk_route(-1, 2, ne, 2, ne, world, r, r * per, (r + 1) * per) == 0);
        EXPECT(dist_chunk_route(q, 1, ne, 2, ne, world, r, r * per, (r + 1) * per) == 0);
        EXPECT(dist_chunk_route(q, 2, ne + 1, 2, ne + 1, world, r, r * per, (r + 1) * per) == 0 || world == 1);
        EXPECT(dist_chunk_route(q + 3, 2, ne, 2, ne, world, r, r * per, (r + 1) * per) == 0);   // fewer than one chunk per rank / not whole
      }
    }
}

int main() {
  test_cr_plans();
  test_lds_cap();
  test_row_blocks();
  test_tile_subsets();
  test_lane_ranges();
  test_chunk_route();
  if (failures) {
    std::fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  std::puts("host_plan OK");
  return 0;
}
