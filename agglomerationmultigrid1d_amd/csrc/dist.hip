// Element-partitioned V-cycle inside the library (BASELINE config 4): one rank's share
// of multigrid_v_cycle (src/solvers.jl:19-50) on its local hierarchy (owned elements + ghost layers),
// with the interface exchanges issued from C++ on the context's stream -- no host language in the
// per-cycle path.  The reference is single-process; there is nothing to mirror here but the cycle.
//
// Schedule (the one agglomerationmultigrid1d_amd/distributed.py documents and the gloo tests pin
// bitwise against the single-GPU cycle): ghost layers deep enough that one V(nPre, nPost) cycle needs
//   1. an all-gather of the interface elements of x0 (W_0 elements per side and rank),
//   2. the coarsest solve across ranks: every rank eliminates the chunks of its own block range of
//      the cyclic reduction, the ranks all-gather their slices of the chunk-boundary system, each
//      solves it and back-substitutes its own chunks, then the coarse ghost blocks are all-gathered
//      (small coarsest levels: all-gather of the owned right-hand side + replicated solve),
// and nothing else: block-Jacobi sweeps are recomputed in the ghost layers.
// Collectives: RCCL (loaded with dlopen so that libaggmg_hip.so carries no link-time dependency on it and
// shares the copy torch has already loaded), or caller-supplied functions (tests drive the same C++ schedule
// over gloo), or a device-local loop-back (rehearsals).
//
// What travels how (round 3: fewer dependent launches per cycle -- every launch of a rank's 1/8 share costs
// about as much in dispatch as in work):
//   * interface elements -> the neighbours' ghost elements: ONE grouped ncclSend / ncclRecv per exchange, straight
//     between the owned interface and the ghost region of the vectors themselves (no pack, no all-gather of
//     every rank's interface to every rank, no unpack).  AGGMG_DIST_P2P=0 brings back pack -> ncclAllGather ->
//     unpack (the form BASELINE.json's north_star words; SURVEY.md 8e names the grouped send/recv as its
//     equivalent).
//   * the chunk-boundary system of the coarsest solve: ncclAllGather IN PLACE -- the chunk kernels write
//     their boundary rows chunk-interleaved, so a rank's chunks are one contiguous slice of the buffer the
//     boundary solve reads as it lies (aggmg_hip.hip, coarse_*_interleaved).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "internal.hpp"

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl_api(std::string* err) {
  static RcclApi api;
  static bool tried = false;
  static std::string why;
  if (!tried) {
    tried = true;
    // the copy already in the process (torch's) if there is one, the ROCm one otherwise; AGGMG_RCCL_LIB names
    // another file (tests use it to take the "not loadable" route)
    const char* forced = getenv("AGGMG_RCCL_LIB");
    std::string last;
    for (const char* name : {forced ? forced : "librccl.so.1", forced ? forced : "librccl.so"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (!api.lib) api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
      const char* e = dlerror();   // read ONCE: the call clears the message
      last = e ? e : (std::string(name) + " not found");
    }
    if (!api.lib) {
      why = std::string("RCCL not loadable: ") + last;
    } else {
      auto sym = [&](const char* n) -> void* {
        void* p = dlsym(api.lib, n);
        if (!p && why.empty()) why = std::string("RCCL symbol missing: ") + n;
        return p;
      };
      api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
      api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
      api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
      api.Send = (decltype(api.Send))sym("ncclSend");
      api.Recv = (decltype(api.Recv))sym("ncclRecv");
      api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
      api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
      api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    }
  }
  if (!why.empty()) {
    if (err) *err = why;
    return nullptr;
  }
  return &api;
}

// out[r * count + i] = in[i] for every rank slot r (loop-back "all-gather" of rehearsals)
__global__ __launch_bounds__(kThreads) void loopback_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                            int64_t count, int world) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= count) return;
  const double v = in[i];
  for (int r = 0; r < world; ++r) out[(int64_t)r * count + i] = v;
}

}  // namespace

struct DevBuf {
  double* p = nullptr;
  int64_t n = 0;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(aggmg_ctx* ctx, int64_t len) {
    n = len;
    HIPCHK(hipMalloc((void**)&p, (size_t)std::max<int64_t>(len, 1) * sizeof(double)));
    HIPCHK(hipMemsetAsync(p, 0, (size_t)std::max<int64_t>(len, 1) * sizeof(double), ctx->stream));
    return AGGMG_OK;
  }
};

// which entries of a local level vector travel in an interface exchange: `send` packs slices of the vector
// into this rank's part of the all-gather (count doubles per rank), `left` / `right` unpack slices of the
// left / right neighbour's part into the ghost entries
struct ExSeg {
  int64_t src, dst, len;
};
struct ExLayout {
  int64_t count = 0;
  std::vector<ExSeg> send, left, right;
};
// the same exchange as neighbour messages: slices [off, off + len) of the local level vector that go to / are
// filled from the left and the right neighbour; rank r's to_right matches rank r + 1's from_left slice by slice
// (equal lengths, same order), to_left matches rank r - 1's from_right
struct NbSeg {
  int64_t off, len;
};
struct NbLayout {
  bool set = false;
  std::vector<NbSeg> to_left, to_right, from_left, from_right;
};

struct aggmg_dist {
  aggmg_hier* H = nullptr;   // local hierarchy, AGGMG_COARSE_EXTERNAL
  aggmg_hier* Hc = nullptr;  // one-level hierarchy of the GLOBAL coarsest operator (replicated)
  int world = 1, rank = 0, nl = 0;
  std::vector<int64_t> own_lo, own_hi, loc_lo, loc_hi, ne;
  std::vector<int> m, W;
  DevBuf send[2], recv[2];  // [0]: finest level, [1]: coarsest level ghosts
  ExLayout ex[2];
  NbLayout nb[2];
  bool p2p = true;          // neighbour messages where the backend has them (AGGMG_DIST_P2P=0: all-gathers only)
  DevBuf rhs_g, sol_g, zero_g;
  bool chunked = false;
  int q = 0;
  int64_t nq = 0, cnt = 0;
  DevBuf zbuf, xq;          // chunk-interleaved boundary rows (one pad block in front), boundary solution
  // collectives
  int backend = 0;  // 0 none, 1 callback, 2 RCCL, 3 loop-back
  aggmg_allgather_fn fn = nullptr;
  aggmg_sendrecv_fn fn_p2p = nullptr;
  void* user = nullptr;
  ncclComm_t comm = nullptr;
  // a communicator of its own for the side stream: two operations of ONE communicator in flight on two streams may
  // start in a different order on different ranks and wait for each other; with one communicator per stream every
  // communicator sees its operations in one order on every rank
  ncclComm_t comm_side = nullptr;
  // interface exchange of the next cycle's x0 under the fine-level ascent
  hipStream_t side = nullptr;
  hipEvent_t ev_main = nullptr, ev_ends = nullptr, ev_side = nullptr, ev_coarse = nullptr;
  bool coarse_overlap = [] { const char* e = getenv("AGGMG_DIST_COARSE_OVERLAP"); return e && e[0] == '1'; }();
  const double* pending = nullptr;
  bool pending_in_place = false;
  int64_t exchanges = 0;
  // hipGraph replay of whole cycles, keyed by the argument tuple (AGGMG_DIST_GRAPH)
  struct GraphKey {
    const double *x0, *b;
    double* x_out;
    int nPre, nPost, flags;
    double alpha;
    bool operator==(const GraphKey& o) const {
      return x0 == o.x0 && b == o.b && x_out == o.x_out && nPre == o.nPre && nPost == o.nPost && flags == o.flags && alpha == o.alpha;
    }
  };
  struct GraphEntry {
    GraphKey key;
    int seen = 0;               // eager runs with this key so far (the first one warms every lazy allocation)
    hipGraphExec_t exec = nullptr;
    int64_t exchanges = 0;      // all-gathers one replay stands for
  };
  std::vector<GraphEntry> graphs;
  bool graph_broken = false;    // a capture failed once (e.g. the collective backend cannot be captured): stay eager
  int64_t graph_replays = 0;
  ~aggmg_dist() {
    for (ncclComm_t c : {comm, comm_side})
      if (c) {
        if (RcclApi* a = rccl_api(nullptr)) (void)a->CommDestroy(c);
      }
    for (auto& g : graphs)
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
    for (hipEvent_t e : {ev_main, ev_ends, ev_side, ev_coarse})
      if (e) (void)hipEventDestroy(e);
    if (side) (void)hipStreamDestroy(side);
  }
};

// the communicator of the stream the call is issued on
static ncclComm_t comm_for(const aggmg_ctx* ctx, const aggmg_dist* d) {
  return (ctx->stream == d->side && d->comm_side) ? d->comm_side : d->comm;
}

static int dist_allgather(aggmg_ctx* ctx, aggmg_dist* d, const double* send, double* recv, int64_t count) {
  d->exchanges += 1;
  if (d->world == 1 && d->backend != 2) {
    if (recv != send) HIPCHK(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return AGGMG_OK;
  }
  switch (d->backend) {
    case 1: {
      const int st = d->fn(d->user, send, recv, count, (void*)ctx->stream);
      if (st != 0) return fail(ctx, AGGMG_ERR_HIP, "aggmg_dist: the all-gather callback reported failure " + std::to_string(st));
      return AGGMG_OK;
    }
    case 2: {
      RcclApi* a = rccl_api(nullptr);
      const ncclResult_t r = a->AllGather(send, recv, (size_t)count, ncclDouble, comm_for(ctx, d), ctx->stream);
      if (r != ncclSuccess) return fail(ctx, AGGMG_ERR_HIP, std::string("ncclAllGather: ") + a->GetErrorString(r));
      return AGGMG_OK;
    }
    case 3: {
      const unsigned nb = (unsigned)((count + kThreads - 1) / kThreads);
      if (nb) hipLaunchKernelGGL(loopback_kernel, dim3(nb), dim3(kThreads), 0, ctx->stream, send, recv, count, d->world);
      HIPCHK(hipGetLastError());
      return AGGMG_OK;
    }
  }
  return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist: no collective backend set (aggmg_dist_init_rccl / aggmg_dist_set_allgather)");
}

static int copy2(aggmg_ctx* ctx, int nseg, const double* const* src, double* const* dst, const int64_t* rows,
                 const int64_t* cols, const int64_t* sld, const int64_t* dld) {
  return aggmg_copy_segments_dev(ctx, nseg, src, dst, rows, cols, sld, dld);
}

// up to four slices per launch (aggmg_copy_segments_dev)
static int copy_segs(aggmg_ctx* ctx, const std::vector<ExSeg>& segs, const double* src_base, double* dst_base) {
  for (size_t i = 0; i < segs.size(); i += 4) {
    const double* src[4];
    double* dst[4];
    int64_t rows[4], cols[4], ld[4];
    int n = 0;
    for (size_t j = i; j < segs.size() && n < 4; ++j) {
      if (segs[j].len <= 0) continue;
      src[n] = src_base + segs[j].src;
      dst[n] = dst_base + segs[j].dst;
      rows[n] = 1, cols[n] = segs[j].len, ld[n] = segs[j].len;
      ++n;
    }
    if (n) CHECK(copy2(ctx, n, src, dst, rows, cols, ld, ld));
  }
  return AGGMG_OK;
}

// the default layout of a level whose DoFs are contiguous per element (DG / agglomerated levels): the first
// and last W owned elements travel, the neighbours' go into the ghost elements
static ExLayout contiguous_layout(const aggmg_dist* d, int level) {
  ExLayout L;
  const int64_t m = d->m[level], wm = (int64_t)d->W[level] * m;
  const int64_t gl = d->own_lo[level] - d->loc_lo[level], gr = d->loc_hi[level] - d->own_hi[level];
  const int64_t o0 = gl * m;
  const int64_t o1 = o0 + (d->own_hi[level] - d->own_lo[level]) * m;
  L.count = 2 * wm;
  L.send = {ExSeg{o0, 0, wm}, ExSeg{o1 - wm, wm, wm}};
  if (gl) L.left = {ExSeg{wm, 0, gl * m}};    // the left neighbour's last W elements
  if (gr) L.right = {ExSeg{0, o1, gr * m}};   // the right neighbour's first W elements
  return L;
}

static int pack_interface(aggmg_ctx* ctx, aggmg_dist* d, const double* x, int level, double* send) {
  return copy_segs(ctx, d->ex[level == 0 ? 0 : 1].send, x, send);
}

// neighbours' interface entries -> ghost entries
static int unpack_ghosts(aggmg_ctx* ctx, aggmg_dist* d, double* x, int level, const double* recv) {
  const ExLayout& L = d->ex[level == 0 ? 0 : 1];
  const int r = d->rank;
  if (r > 0 && !L.left.empty()) CHECK(copy_segs(ctx, L.left, recv + (int64_t)(r - 1) * L.count, x));
  if (r + 1 < d->world && !L.right.empty()) CHECK(copy_segs(ctx, L.right, recv + (int64_t)(r + 1) * L.count, x));
  return AGGMG_OK;
}

static NbLayout contiguous_neighbors(const aggmg_dist* d, int level) {
  NbLayout L;
  const int64_t m = d->m[level], wm = (int64_t)d->W[level] * m;
  const int64_t gl = d->own_lo[level] - d->loc_lo[level], gr = d->loc_hi[level] - d->own_hi[level];
  const int64_t o0 = gl * m;
  const int64_t o1 = o0 + (d->own_hi[level] - d->own_lo[level]) * m;
  L.set = true;
  if (wm > 0) {
    L.to_left = {NbSeg{o0, wm}};
    L.to_right = {NbSeg{o1 - wm, wm}};
    if (gl) L.from_left = {NbSeg{0, gl * m}};
    if (gr) L.from_right = {NbSeg{o1, gr * m}};
  }
  return L;
}

static bool use_p2p(const aggmg_dist* d, int slot) {
  if (!d->p2p || !d->nb[slot].set) return false;
  return d->backend == 2 || d->backend == 3 || (d->backend == 1 && d->fn_p2p);
}

// interface slices of x -> the neighbours' ghost slices of THEIR x, theirs -> mine; in place, one launch
static int neighbor_exchange(aggmg_ctx* ctx, aggmg_dist* d, double* x, int slot) {
  const NbLayout& L = d->nb[slot];
  const int r = d->rank;
  const bool hasL = r > 0, hasR = r + 1 < d->world;
  d->exchanges += 1;
  switch (d->backend) {
    case 2: {
      RcclApi* a = rccl_api(nullptr);
      const ncclComm_t comm = comm_for(ctx, d);
      ncclResult_t st = a->GroupStart();
      auto post = [&](const std::vector<NbSeg>& out, const std::vector<NbSeg>& in, int peer) {
        for (const NbSeg& g : out)
          if (st == ncclSuccess && g.len > 0) st = a->Send(x + g.off, (size_t)g.len, ncclDouble, peer, comm, ctx->stream);
        for (const NbSeg& g : in)
          if (st == ncclSuccess && g.len > 0) st = a->Recv(x + g.off, (size_t)g.len, ncclDouble, peer, comm, ctx->stream);
      };
      if (hasL) post(L.to_left, L.from_left, r - 1);
      if (hasR) post(L.to_right, L.from_right, r + 1);
      const ncclResult_t en = a->GroupEnd();
      if (st == ncclSuccess) st = en;
      if (st != ncclSuccess) return fail(ctx, AGGMG_ERR_HIP, std::string("ncclSend / ncclRecv: ") + a->GetErrorString(st));
      return AGGMG_OK;
    }
    case 1: {
      std::vector<int> peer, is_send;
      std::vector<double*> ptr;
      std::vector<int64_t> cnt;
      auto post = [&](const std::vector<NbSeg>& out, const std::vector<NbSeg>& in, int pr) {
        for (const NbSeg& g : out)
          if (g.len > 0) peer.push_back(pr), is_send.push_back(1), ptr.push_back(x + g.off), cnt.push_back(g.len);
        for (const NbSeg& g : in)
          if (g.len > 0) peer.push_back(pr), is_send.push_back(0), ptr.push_back(x + g.off), cnt.push_back(g.len);
      };
      if (hasL) post(L.to_left, L.from_left, r - 1);
      if (hasR) post(L.to_right, L.from_right, r + 1);
      if (peer.empty()) return AGGMG_OK;
      const int st = d->fn_p2p(d->user, (int)peer.size(), peer.data(), is_send.data(), ptr.data(), cnt.data(), (void*)ctx->stream);
      if (st != 0) return fail(ctx, AGGMG_ERR_HIP, "aggmg_dist: the send/recv callback reported failure " + std::to_string(st));
      return AGGMG_OK;
    }
    case 3: {
      // rehearsal: this rank's own interface stands in for the neighbours' (same bytes, one launch like the
      // grouped RCCL call it replaces)
      const double* src[4];
      double* dst[4];
      int64_t rows[4], cols[4], ld[4];
      int n = 0;
      auto pair = [&](const std::vector<NbSeg>& out, const std::vector<NbSeg>& in) {
        for (size_t i = 0; i < in.size() && i < out.size() && n < 4; ++i) {
          if (in[i].len <= 0) continue;
          src[n] = x + out[i].off, dst[n] = x + in[i].off;
          rows[n] = 1, cols[n] = std::min(in[i].len, out[i].len), ld[n] = cols[n];
          ++n;
        }
      };
      if (hasL) pair(L.to_right, L.from_left);    // what a left neighbour like this rank would send
      if (hasR) pair(L.to_left, L.from_right);
      if (n) CHECK(copy2(ctx, n, src, dst, rows, cols, ld, ld));
      return AGGMG_OK;
    }
  }
  return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist: no collective backend set (aggmg_dist_init_rccl / aggmg_dist_set_allgather)");
}

static int exchange_ghosts(aggmg_ctx* ctx, aggmg_dist* d, double* x, int level) {
  const int slot = level == 0 ? 0 : 1;
  if (d->world == 1 || d->ex[slot].count == 0) return AGGMG_OK;
  if (use_p2p(d, slot)) return neighbor_exchange(ctx, d, x, slot);
  CHECK(pack_interface(ctx, d, x, level, d->send[slot].p));
  CHECK(dist_allgather(ctx, d, d->send[slot].p, d->recv[slot].p, d->ex[slot].count));
  return unpack_ghosts(ctx, d, x, level, d->recv[slot].p);
}

extern "C" int aggmg_dist_create(aggmg_ctx* ctx, aggmg_hier* local, aggmg_hier* coarse_global, int world, int rank,
                                 int nlevels, const int64_t* own_lo, const int64_t* own_hi, const int64_t* loc_lo,
                                 const int64_t* loc_hi, const int64_t* ne, const int32_t* m, const int32_t* W,
                                 aggmg_dist** out) {
  if (!ctx || !out) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_create: NULL argument");
  *out = nullptr;
  if (!local || !coarse_global || !own_lo || !own_hi || !loc_lo || !loc_hi || !ne || !m || !W)
    return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_create: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_create: bad rank / world");
  if (nlevels != (int)local->lv.size() || nlevels < 2)
    return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_dist_create: nlevels does not match the local hierarchy (>= 2 levels)");
  if (local->coarse_mode != AGGMG_COARSE_EXTERNAL)
    return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_create: the local hierarchy must be created with AGGMG_COARSE_EXTERNAL");
  if (coarse_global->lv.size() != 1) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_create: the coarse hierarchy must have one level");
  HIPCHK(hipSetDevice(ctx->device));
  std::unique_ptr<aggmg_dist> d(new aggmg_dist());
  d->H = local;
  d->Hc = coarse_global;
  d->world = world;
  d->rank = rank;
  d->nl = nlevels;
  for (int k = 0; k < nlevels; ++k) {
    if (m[k] < 1 || W[k] < 0 || !(0 <= loc_lo[k] && loc_lo[k] <= own_lo[k] && own_lo[k] <= own_hi[k] &&
                                   own_hi[k] <= loc_hi[k] && loc_hi[k] <= ne[k]))
      return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_create: inconsistent element ranges at level " + std::to_string(k + 1));
    // (a CG level holds one more DoF than elements x m: the last vertex)
    if ((loc_hi[k] - loc_lo[k]) * m[k] != local->lv[k].N && (loc_hi[k] - loc_lo[k]) * m[k] + 1 != local->lv[k].N)
      return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_dist_create: local level size does not match the layout at level " + std::to_string(k + 1));
    const int64_t gl = own_lo[k] - loc_lo[k], gr = loc_hi[k] - own_hi[k];
    if ((gl && gl != W[k]) || (gr && gr != W[k]) || (world > 1 && own_hi[k] - own_lo[k] < W[k]))
      return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_create: ghost layers must be W elements wide (or absent at a domain end)");
    d->own_lo.push_back(own_lo[k]), d->own_hi.push_back(own_hi[k]);
    d->loc_lo.push_back(loc_lo[k]), d->loc_hi.push_back(loc_hi[k]);
    d->ne.push_back(ne[k]), d->m.push_back(m[k]), d->W.push_back(W[k]);
  }
  const int nc = nlevels - 1;
  if (ne[nc] * m[nc] != coarse_global->lv[0].N)
    return fail(ctx, AGGMG_ERR_DIMENSION, "aggmg_dist_create: global coarsest operator size does not match the layout");
  for (int s = 0; s < 2; ++s) {
    const int lev = s == 0 ? 0 : nc;
    d->ex[s] = contiguous_layout(d.get(), lev);
    d->nb[s] = contiguous_neighbors(d.get(), lev);
    CHECK(d->send[s].alloc(ctx, d->ex[s].count));
    CHECK(d->recv[s].alloc(ctx, d->ex[s].count * world));
  }
  {
    const char* e = getenv("AGGMG_DIST_P2P");
    d->p2p = !(e && e[0] == '0');
  }
  const int64_t Ng = ne[nc] * m[nc];
  CHECK(d->rhs_g.alloc(ctx, Ng));
  CHECK(d->sol_g.alloc(ctx, Ng));
  CHECK(d->zero_g.alloc(ctx, Ng));
  // chunked coarsest solve when the replicated solver has a chunk plan that the partition respects
  if (world > 1) {
    int q = -1, mblk = 0;
    int64_t nq = 0, nblk = 0;
    CHECK(aggmg_coarse_plan(ctx, coarse_global, &q, &nq, &mblk, &nblk));
    const int64_t own_blk = own_hi[nc] - own_lo[nc];
    // every rank must own the same number of whole chunks, rank r the r-th run of them: the in-place all-gather
    // has equal counts and puts rank r's slice at r * count.  The decision must come out the same on every rank
    // (mismatched collective counts hang), so it is taken from GLOBAL quantities only -- the replicated operator's
    // plan, the level's block count and the world size; a rank whose own range then breaks the pattern is refused
    // here instead of taking another route than its peers.
    const int route = dist_chunk_route(q, mblk, nblk, m[nc], ne[nc], world, rank, own_lo[nc], own_hi[nc]);   // host_plan.hpp
    if (route < 0)
      return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_create: the coarsest level divides evenly into whole chunks per rank, so every "
                                           "rank must own blocks [rank * n / world, (rank + 1) * n / world)");
    if (route == 1) {
      d->chunked = true;
      d->q = q;
      d->nq = nq;
      d->cnt = (own_blk >> q) * mblk;
      CHECK(d->zbuf.alloc(ctx, (nq + 2) * 2 * mblk));   // zeroed: the pad block in front stays zero
      CHECK(d->xq.alloc(ctx, (nq + 1) * mblk));
    }
  }
  HIPCHK(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
  for (hipEvent_t* e : {&d->ev_main, &d->ev_ends, &d->ev_side, &d->ev_coarse}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *out = d.release();
  return AGGMG_OK;
}

extern "C" int aggmg_dist_free(aggmg_ctx* ctx, aggmg_dist* d) {
  if (!ctx) return AGGMG_ERR_ARGUMENT;
  if (!d) return AGGMG_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (d->side) HIPCHK(hipStreamSynchronize(d->side));
  delete d;
  return AGGMG_OK;
}

// Levels whose DoFs are not contiguous per element (CG levels in the reference's vertices-first numbering:
// the interface is a slice of the vertex part plus a slice of the interior part) describe their exchange
// explicitly: `count` doubles per rank; send_*: slices of the local vector packed into this rank's part;
// left_* / right_*: slices of the left / right neighbour's part unpacked into the ghost entries.
extern "C" int aggmg_dist_set_exchange_layout(aggmg_ctx* ctx, aggmg_dist* d, int level, int64_t count, int nsend,
                                              const int64_t* send_src, const int64_t* send_dst, const int64_t* send_len,
                                              int nleft, const int64_t* left_src, const int64_t* left_dst,
                                              const int64_t* left_len, int nright, const int64_t* right_src,
                                              const int64_t* right_dst, const int64_t* right_len) {
  if (!ctx || !d) return AGGMG_ERR_ARGUMENT;
  if (level != 0 && level != d->nl - 1)
    return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_set_exchange_layout: only the finest and the coarsest level exchange ghosts");
  if (count < 0 || nsend < 0 || nleft < 0 || nright < 0)
    return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_set_exchange_layout: negative count");
  const int slot = level == 0 ? 0 : 1;
  const int64_t N = d->H->lv[level].N;
  ExLayout L;
  L.count = count;
  auto fill = [&](std::vector<ExSeg>& v, int n, const int64_t* a, const int64_t* b, const int64_t* len, int64_t lim_src,
                  int64_t lim_dst) -> bool {
    for (int i = 0; i < n; ++i) {
      if (len[i] < 0 || a[i] < 0 || b[i] < 0 || a[i] + len[i] > lim_src || b[i] + len[i] > lim_dst) return false;
      v.push_back(ExSeg{a[i], b[i], len[i]});
    }
    return true;
  };
  if (!fill(L.send, nsend, send_src, send_dst, send_len, N, count) || !fill(L.left, nleft, left_src, left_dst, left_len, count, N) ||
      !fill(L.right, nright, right_src, right_dst, right_len, count, N))
    return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_set_exchange_layout: a slice leaves the vector or the exchange buffer");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  d->ex[slot] = L;
  d->nb[slot] = NbLayout();   // the contiguous default no longer describes this level: aggmg_dist_set_neighbor_layout
  d->send[slot].~DevBuf();
  new (&d->send[slot]) DevBuf();
  d->recv[slot].~DevBuf();
  new (&d->recv[slot]) DevBuf();
  CHECK(d->send[slot].alloc(ctx, count));
  CHECK(d->recv[slot].alloc(ctx, count * d->world));
  for (auto& g : d->graphs)   // captured cycles refer to the old buffers
    if (g.exec) {
      (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
      g.seen = 0;
    }
  return AGGMG_OK;
}

extern "C" int aggmg_dist_set_neighbor_layout(aggmg_ctx* ctx, aggmg_dist* d, int level, int nto_left, const int64_t* to_left_off,
                                              const int64_t* to_left_len, int nto_right, const int64_t* to_right_off,
                                              const int64_t* to_right_len, int nfrom_left, const int64_t* from_left_off,
                                              const int64_t* from_left_len, int nfrom_right, const int64_t* from_right_off,
                                              const int64_t* from_right_len) {
  if (!ctx || !d) return AGGMG_ERR_ARGUMENT;
  if (level != 0 && level != d->nl - 1)
    return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_set_neighbor_layout: only the finest and the coarsest level exchange ghosts");
  const int64_t N = d->H->lv[level].N;
  NbLayout L;
  auto fill = [&](std::vector<NbSeg>& v, int n, const int64_t* off, const int64_t* len) -> bool {
    if (n < 0 || n > 2 || (n && (!off || !len))) return false;
    for (int i = 0; i < n; ++i) {
      if (off[i] < 0 || len[i] < 0 || off[i] + len[i] > N) return false;
      v.push_back(NbSeg{off[i], len[i]});
    }
    return true;
  };
  if (!fill(L.to_left, nto_left, to_left_off, to_left_len) || !fill(L.to_right, nto_right, to_right_off, to_right_len) ||
      !fill(L.from_left, nfrom_left, from_left_off, from_left_len) || !fill(L.from_right, nfrom_right, from_right_off, from_right_len))
    return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_set_neighbor_layout: at most two slices per direction, inside the level vector");
  L.set = true;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  d->nb[level == 0 ? 0 : 1] = L;
  for (auto& g : d->graphs)
    if (g.exec) {
      (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
      g.seen = 0;
    }
  return AGGMG_OK;
}

extern "C" int aggmg_dist_set_sendrecv(aggmg_ctx* ctx, aggmg_dist* d, aggmg_sendrecv_fn fn) {
  if (!ctx || !d) return AGGMG_ERR_ARGUMENT;
  d->fn_p2p = fn;
  return AGGMG_OK;
}

extern "C" int aggmg_dist_set_allgather(aggmg_ctx* ctx, aggmg_dist* d, aggmg_allgather_fn fn, void* user) {
  if (!ctx || !d) return AGGMG_ERR_ARGUMENT;
  d->fn = fn;
  d->user = user;
  d->backend = fn ? 1 : 0;
  return AGGMG_OK;
}

extern "C" int aggmg_dist_set_loopback(aggmg_ctx* ctx, aggmg_dist* d) {
  if (!ctx || !d) return AGGMG_ERR_ARGUMENT;
  d->backend = 3;
  return AGGMG_OK;
}

extern "C" int aggmg_rccl_available(char* why_out, int nbytes) {
  std::string why;
  RcclApi* a = rccl_api(&why);
  if (why_out && nbytes > 0) {
    const size_t n = std::min<size_t>(why.size(), (size_t)nbytes - 1);
    std::memcpy(why_out, why.data(), n);
    why_out[n] = 0;
  }
  return a ? AGGMG_OK : AGGMG_ERR_UNSUPPORTED;
}

extern "C" int aggmg_rccl_unique_id(aggmg_ctx* ctx, void* id_out, int nbytes) {
  if (!ctx || !id_out) return AGGMG_ERR_ARGUMENT;
  if (nbytes < (int)sizeof(ncclUniqueId)) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_rccl_unique_id: buffer smaller than AGGMG_RCCL_ID_BYTES");
  std::string why;
  RcclApi* a = rccl_api(&why);
  if (!a) return fail(ctx, AGGMG_ERR_UNSUPPORTED, why);
  ncclUniqueId id;
  const ncclResult_t r = a->GetUniqueId(&id);
  if (r != ncclSuccess) return fail(ctx, AGGMG_ERR_HIP, std::string("ncclGetUniqueId: ") + a->GetErrorString(r));
  std::memcpy(id_out, &id, sizeof(id));
  return AGGMG_OK;
}

extern "C" int aggmg_dist_init_rccl(aggmg_ctx* ctx, aggmg_dist* d, const void* id, int nbytes, int* nranks_out) {
  if (!ctx || !d || !id) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_init_rccl: NULL argument");
  if (nbytes < (int)sizeof(ncclUniqueId)) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_init_rccl: id smaller than AGGMG_RCCL_ID_BYTES");
  std::string why;
  RcclApi* a = rccl_api(&why);
  if (!a) return fail(ctx, AGGMG_ERR_UNSUPPORTED, why);
  HIPCHK(hipSetDevice(ctx->device));
  // one id: one communicator; two ids back to back (2 * AGGMG_RCCL_ID_BYTES): a second communicator for the side stream
  const int nid = nbytes >= 2 * AGGMG_RCCL_ID_BYTES ? 2 : 1;
  ncclComm_t made[2] = {nullptr, nullptr};
  int cnt = 0;
  for (int k = 0; k < nid; ++k) {
    ncclUniqueId uid;
    std::memcpy(&uid, static_cast<const char*>(id) + (size_t)k * AGGMG_RCCL_ID_BYTES, sizeof(uid));
    ncclResult_t r = a->CommInitRank(&made[k], d->world, uid, d->rank);
    if (r == ncclSuccess) r = a->CommCount(made[k], &cnt);
    if (r != ncclSuccess || cnt != d->world) {
      for (ncclComm_t c : made)
        if (c) (void)a->CommDestroy(c);
      if (r != ncclSuccess) return fail(ctx, AGGMG_ERR_HIP, std::string("ncclCommInitRank: ") + a->GetErrorString(r));
      return fail(ctx, AGGMG_ERR_HIP, "aggmg_dist_init_rccl: communicator reports " + std::to_string(cnt) + " ranks, expected " +
                                          std::to_string(d->world));
    }
  }
  for (ncclComm_t c : {d->comm, d->comm_side})
    if (c) (void)a->CommDestroy(c);
  d->comm = made[0];
  d->comm_side = made[1];
  d->backend = 2;
  if (nranks_out) *nranks_out = cnt;
  return AGGMG_OK;
}

extern "C" int aggmg_dist_allgather_dev(aggmg_ctx* ctx, aggmg_dist* d, const double* send, double* recv, int64_t count) {
  if (!ctx || !d || !send || !recv || count < 0) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_allgather_dev: bad argument");
  return dist_allgather(ctx, d, send, recv, count);
}

extern "C" int aggmg_dist_exchange_ghosts_dev(aggmg_ctx* ctx, aggmg_dist* d, double* x_local, int level) {
  if (!ctx || !d || !x_local) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_exchange_ghosts_dev: NULL argument");
  if (level != 0 && level != d->nl - 1)
    return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_exchange_ghosts_dev: only the finest and the coarsest level exchange ghosts");
  return exchange_ghosts(ctx, d, x_local, level);
}

extern "C" int aggmg_dist_set_coarse_overlap(aggmg_ctx* ctx, aggmg_dist* d, int on) {
  if (!ctx || !d) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_set_coarse_overlap: NULL argument");
  d->coarse_overlap = on != 0;
  for (auto& g : d->graphs)   // captured cycles hold the schedule they were captured with
    if (g.exec) {
      (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
      g.seen = 0;
    }
  return AGGMG_OK;
}

extern "C" int aggmg_dist_info(aggmg_ctx* ctx, const aggmg_dist* d, int64_t* exchanges, int* chunked, int* backend) {
  if (!ctx || !d) return AGGMG_ERR_ARGUMENT;
  if (exchanges) *exchanges = d->exchanges;
  if (chunked) *chunked = d->chunked ? 1 : 0;
  if (backend) *backend = d->backend;
  return AGGMG_OK;
}

// one cycle, issued launch by launch on ctx->stream (and the side stream for the overlapped exchange,
// which is joined back before returning: the cycle is self-contained, hence capturable)
static int dist_vcycle_eager(aggmg_ctx* ctx, aggmg_dist* d, double* x0, const double* b, double* x_out, int nPre,
                             int nPost, double alpha, int flags) {
  aggmg_hier* H = d->H;
  const int nc = d->nl - 1;
  const int mc = d->m[nc];
  bool ghosts_valid = (flags & AGGMG_DIST_X0_GHOSTS_VALID) != 0;
  bool coarse_done = false;   // the levels below the finest have been run already (split around the coarse ghost exchange)
  hipStream_t main = ctx->stream;
  if (d->pending) {
    // the exchange issued under the previous cycle's ascent (already joined to the main stream): use
    // it if it was for this x0
    if (d->pending == x0) {
      if (!d->pending_in_place) CHECK(unpack_ghosts(ctx, d, x0, 0, d->recv[0].p));   // (neighbour messages landed in the ghosts)
      ghosts_valid = true;
    }
    d->pending = nullptr;
  }
  if (!ghosts_valid) CHECK(exchange_ghosts(ctx, d, x0, 0));
  CHECK(aggmg_vcycle_down_dev(ctx, H, x0, b, nPre, alpha));
  double* rhs_c = H->lv[nc].rhs;
  double* sol_c = H->lv[nc].u[0];
  const int64_t gl = d->own_lo[nc] - d->loc_lo[nc];
  const int64_t own_c = (d->own_hi[nc] - d->own_lo[nc]) * mc;
  const double* own = rhs_c + gl * mc;
  if (d->chunked) {
    const int64_t blo = d->own_lo[nc], bhi = d->own_hi[nc];
    const int64_t clo = blo >> d->q;
    const int64_t cnt = d->cnt;
    // the chunk kernels write their boundary rows chunk-interleaved at GLOBAL chunk positions of Z (one pad block
    // in front): this rank's chunks are the contiguous slice Z[clo .. clo + chunks), 2 cnt doubles, and the
    // all-gather runs in place -- no pack, no unpack, the boundary solve reads Z as it lies
    double* Z = d->zbuf.p + 2 * mc;
    CHECK(coarse_chunk_forward_interleaved(ctx, d->Hc, own, blo, bhi, Z));
    CHECK(dist_allgather(ctx, d, Z + clo * 2 * mc, Z, 2 * cnt));
    CHECK(coarse_boundary_solve_interleaved(ctx, d->Hc, Z, d->xq.p));
    CHECK(aggmg_coarse_chunk_backward_dev(ctx, d->Hc, own, blo, bhi, d->xq.p, sol_c + gl * mc));
    // The neighbours' ghost blocks of the coarsest solution are read by the end tiles of the ascent only: the exchange
    // can go to the side stream and the tiles in between run under it (aggmg_dist_set_coarse_overlap; off by default:
    // with loop-back stand-ins the extra launch and the two stream joins cost 9 us against the 5 us copy they hide, a
    // real neighbour exchange of 15 - 25 us would pay -- bench.py times both in its warm-up and keeps the faster).
    // Needs the levels below the finest as one two-level launch; anything else waits for the exchange.
    if (d->coarse_overlap && d->world > 1 && d->ex[1].count > 0) {
      HIPCHK(hipEventRecord(d->ev_main, main));
      HIPCHK(hipStreamWaitEvent(d->side, d->ev_main, 0));
      ctx->stream = d->side;
      int st = exchange_ghosts(ctx, d, sol_c, nc);
      (void)hipEventRecord(d->ev_coarse, d->side);
      ctx->stream = main;
      CHECK(st);
      const int64_t gh_lo = d->own_lo[nc] - d->loc_lo[nc], gh_hi = d->loc_hi[nc] - d->own_hi[nc];
      st = aggmg_vcycle_up_coarse_dev(ctx, H, b, nPost, alpha, 2, gh_lo, gh_hi);     // the tiles that read no ghost block
      HIPCHK(hipStreamWaitEvent(main, d->ev_coarse, 0));
      if (st == AGGMG_OK) {
        CHECK(aggmg_vcycle_up_coarse_dev(ctx, H, b, nPost, alpha, 1, gh_lo, gh_hi)); // the end tiles
        coarse_done = true;
      } else if (st != AGGMG_ERR_UNSUPPORTED) {
        return st;
      }
    } else {
      CHECK(exchange_ghosts(ctx, d, sol_c, nc));
    }
  } else {
    CHECK(dist_allgather(ctx, d, own, d->rhs_g.p, own_c));
    // a one-level hierarchy's V-cycle IS the coarsest direct solve (src/solvers.jl:39)
    CHECK(aggmg_vcycle_dev(ctx, d->Hc, d->zero_g.p, d->rhs_g.p, 0, 0, 1.0, d->sol_g.p));
    HIPCHK(hipMemcpyAsync(sol_c, d->sol_g.p + d->loc_lo[nc] * mc, (size_t)(d->loc_hi[nc] - d->loc_lo[nc]) * mc * sizeof(double),
                          hipMemcpyDeviceToDevice, ctx->stream));
  }
  const bool overlap = (flags & AGGMG_DIST_OVERLAP_NEXT) && d->world > 1 && d->W[0] > 0;
  if (overlap) {
    const int64_t gl0 = d->own_lo[0] - d->loc_lo[0];
    const int64_t head = gl0 + d->W[0];                                      // local elements [0, head)
    const int64_t tail = gl0 + (d->own_hi[0] - d->own_lo[0]) - d->W[0];      // [tail, end)
    int st = coarse_done ? AGGMG_OK : aggmg_vcycle_up_split_dev(ctx, H, b, nPost, alpha, x_out, head, tail, 0);  // the coarser levels
    if (st == AGGMG_OK) {
      // side stream: the tiles holding the interface elements, then their pack + all-gather
      HIPCHK(hipEventRecord(d->ev_main, main));
      HIPCHK(hipStreamWaitEvent(d->side, d->ev_main, 0));
      ctx->stream = d->side;
      st = aggmg_vcycle_up_split_dev(ctx, H, b, nPost, alpha, x_out, head, tail, 1);
      const bool in_place = use_p2p(d, 0);
      if (st == AGGMG_OK) {
        (void)hipEventRecord(d->ev_ends, d->side);
        // neighbour messages go from x_out's interface elements straight into the neighbours' ghost elements of
        // THEIR x_out (the middle tiles on the main stream write neither)
        st = in_place ? neighbor_exchange(ctx, d, x_out, 0) : pack_interface(ctx, d, x_out, 0, d->send[0].p);
      }
      if (st == AGGMG_OK && !in_place) st = dist_allgather(ctx, d, d->send[0].p, d->recv[0].p, d->ex[0].count);
      (void)hipEventRecord(d->ev_side, d->side);
      ctx->stream = main;
      CHECK(st);
      d->pending = x_out;
      d->pending_in_place = in_place;
      CHECK(aggmg_vcycle_up_split_dev(ctx, H, b, nPost, alpha, x_out, head, tail, 2));  // the middle, main stream
      HIPCHK(hipStreamWaitEvent(main, d->ev_side, 0));  // join: x_out whole and the gathered interface ready
      return AGGMG_OK;
    }
    if (st != AGGMG_ERR_UNSUPPORTED) return st;  // (not a fused block-tridiagonal fine level: plain ascent)
  }
  if (coarse_done) return aggmg_vcycle_up_split_dev(ctx, H, b, nPost, alpha, x_out, 0, 0, 3);   // the finest level alone
  return aggmg_vcycle_up_dev(ctx, H, b, nPost, alpha, x_out);
}

extern "C" int aggmg_dist_vcycle_dev(aggmg_ctx* ctx, aggmg_dist* d, double* x0, const double* b, double* x_out, int nPre,
                                     int nPost, double alpha, int flags) {
  if (!ctx || !d || !x0 || !b || !x_out) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_vcycle_dev: NULL argument");
  if (nPre < 0 || nPost < 0) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_vcycle_dev: negative sweep count");
  if (x_out == x0 || x_out == b) return fail(ctx, AGGMG_ERR_ARGUMENT, "aggmg_dist_vcycle_dev: x_out must not alias x0 or b");
  // hipGraph replay: a cycle with the same arguments was issued eagerly once (lazy allocations done)
  // and captured on its second call; host callbacks, the host coarsest solver and the event profiler
  // cannot be captured
  const bool graphable = (flags & AGGMG_DIST_GRAPH) && !d->graph_broken && ctx->profiling == 0 && d->backend != 1 &&
                         d->Hc->cr.valid;
  if (!graphable) return dist_vcycle_eager(ctx, d, x0, b, x_out, nPre, nPost, alpha, flags);
  // whether this call consumes a prefetched exchange is part of what the graph does
  const int eff = flags | (d->pending == x0 && d->pending ? 0x100 : 0);
  const aggmg_dist::GraphKey key{x0, b, x_out, nPre, nPost, eff, alpha};
  aggmg_dist::GraphEntry* g = nullptr;
  for (auto& e : d->graphs)
    if (e.key == key) g = &e;
  if (!g) {
    if (d->graphs.size() >= 8) return dist_vcycle_eager(ctx, d, x0, b, x_out, nPre, nPost, alpha, flags);
    d->graphs.push_back(aggmg_dist::GraphEntry{key});
    g = &d->graphs.back();
  }
  if (g->exec) {
    HIPCHK(hipGraphLaunch(g->exec, ctx->stream));
    d->exchanges += g->exchanges;
    d->graph_replays += 1;
    d->pending = (flags & AGGMG_DIST_OVERLAP_NEXT) && d->world > 1 && d->W[0] > 0 ? x_out : nullptr;
    d->pending_in_place = use_p2p(d, 0);
    return AGGMG_OK;
  }
  if (g->seen++ == 0) return dist_vcycle_eager(ctx, d, x0, b, x_out, nPre, nPost, alpha, flags);
  // second call with this key: capture it
  const int64_t ex0 = d->exchanges;
  const double* pending0 = d->pending;
  // (the legacy default stream cannot be captured: run the library on its own stream, aggmg_reset_stream)
  hipError_t e = ctx->stream ? hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) : hipErrorStreamCaptureUnsupported;
  if (e != hipSuccess) {
    (void)hipGetLastError();  // the refusal must not surface at the next launch check
    d->graph_broken = true;
    return dist_vcycle_eager(ctx, d, x0, b, x_out, nPre, nPost, alpha, flags);
  }
  const int st = dist_vcycle_eager(ctx, d, x0, b, x_out, nPre, nPost, alpha, flags);
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(ctx->stream, &graph);
  hipGraphExec_t exec = nullptr;
  if (st == AGGMG_OK && e == hipSuccess && graph) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (graph) (void)hipGraphDestroy(graph);
  if (st != AGGMG_OK || e != hipSuccess || !exec) {
    // nothing has run: fall back to issuing this cycle eagerly, and stay eager from now on
    (void)hipGetLastError();
    d->graph_broken = true;
    d->exchanges = ex0;
    d->pending = pending0;
    return dist_vcycle_eager(ctx, d, x0, b, x_out, nPre, nPost, alpha, flags);
  }
  g->exec = exec;
  g->exchanges = d->exchanges - ex0;
  HIPCHK(hipGraphLaunch(exec, ctx->stream));
  d->graph_replays += 1;
  return AGGMG_OK;
}

extern "C" int aggmg_dist_graph_info(aggmg_ctx* ctx, const aggmg_dist* d, int64_t* replays, int* captured, int* broken) {
  if (!ctx || !d) return AGGMG_ERR_ARGUMENT;
  if (replays) *replays = d->graph_replays;
  if (captured) {
    int n = 0;
    for (auto& g : d->graphs) n += g.exec ? 1 : 0;
    *captured = n;
  }
  if (broken) *broken = d->graph_broken ? 1 : 0;
  return AGGMG_OK;
}
