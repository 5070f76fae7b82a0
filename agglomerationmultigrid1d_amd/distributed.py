"""Element-range partition of the V-cycle across the GPUs of one node (BASELINE configs 4 and 5).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI; "gloo" for the tests on
one box).  Rank r owns a contiguous range of fine elements and, level by level, the agglomerates
made of them; it stores its owned elements plus `W_k` ghost elements per side and runs the very
same fused kernels on that local domain.

Communication-avoiding schedule.  Jacobi / block-Jacobi sweeps are order independent and in 1-D a
sweep moves information by exactly one element, so instead of one interface exchange per operator
application (7 per level per cycle) the ghost layers are made deep enough that a whole V(nPre,
nPost) cycle needs THREE small all-gathers:

    1. the interface DoFs of x0 (W_0 elements per side and rank) -- issued under the previous cycle's
       fine-level ascent on a second stream when the caller loops (overlap_next),
    2. the coarsest solve across ranks: every rank eliminates the chunks of its own block range of the
       cyclic reduction, the chunk-boundary system is all-gathered and solved redundantly,
    3. after back substitution, the coarse ghost blocks
       (small coarsest levels: ONE all-gather of the owned right-hand side + replicated solve instead
       of 2 and 3),

everything else being recomputed redundantly in the ghost layers (0.03 % extra work at 2^22
elements on 8 ranks).  `halo_widths` derives the widths from (ratios, nPre, nPost) by tracking
how far validity shrinks: a sweep costs one element per side (two for the red-black Gauss-Seidel
extension), the residual one more, a transfer divides / multiplies by the ratio.  Owned values are
bitwise those of the single-GPU run (same per-row arithmetic, tiles only differ in where they start).

Two drivers of the same schedule:
  * NativeDistributedVCycle -- the schedule inside libaggmg_hip.so (csrc/dist.hip, C ABI aggmg_dist_*):
    one C call per cycle, RCCL all-gathers issued from C++ (or routed back to torch.distributed: the
    gloo tests), optional hipGraph replay.  What bench.py runs.  DG / agglomerated hierarchies
    (RankLayout, build_local_uniform; block Gauss-Seidel included) and CG p-chain hierarchies
    (CgRankLayout, build_local_cg: vertices-first numbering on the rank's sub-mesh, one shared vertex
    per interface).
  * DistributedVCycle -- the same steps in Python against a small engine interface, so that CPU tests
    can run it (`gloo`, world_size 2) with a NumPy engine standing in for the GPU kernels.

No multi-GPU scaling curve has been measured: the build boxes have one GPU.  RCCL with more than one
rank has never run here; the library's RCCL path is exercised with the single rank a box allows
(tests/test_distributed_gpu.py) and the exchange pattern with 2 - 4 ranks over gloo.
"""
import ctypes
import json
import os
import time

import numpy as np


# ------------------------------------------------------------------------------------------
# layout
# ------------------------------------------------------------------------------------------
def halo_widths(ratios, nPre, nPost, gs=False):
    """Ghost elements per side for levels 0..len(ratios) (last = coarsest, held replicated but
    copied locally with W ghosts), nested so that a local level is exactly the children of the
    local next-coarser level: W_k = ratios[k] * W_{k+1}.  Smallest widths for which the owned part
    of the V-cycle result is exact with a single exchange of x0 and the coarsest gather.
    gs: red-black block Gauss-Seidel sweeps (the labelled extension) -- a sweep is two half-sweeps
    and moves information by two elements (one when it starts from a zero iterate)."""
    nl = len(ratios) + 1
    per = 2 if gs else 1
    # (gs: the two colours are the parity of the LOCAL element index, so local ranges must start on even
    # elements: even widths on every level)
    for w in range(0, 4096, 2 if gs else 1):
        W = [0] * nl
        W[nl - 1] = w
        for k in range(nl - 2, -1, -1):
            W[k] = ratios[k] * W[k + 1]
        # descending: validity margin (elements beyond the owned boundary) of the pre-smoothed u
        Pu = [0] * nl
        Rm = W[0]                       # rhs of level 0 = b, valid on the whole local domain
        ok = True
        for k in range(nl - 1):
            # levels below the finest start from u = 0 (src/solvers.jl:29-31): their first sweep
            # reads no neighbour and costs no margin
            Pu[k] = min(Rm, W[k]) - (per * nPre if k == 0 else max(per * nPre - 1, 0))
            if Pu[k] < 1:               # the residual needs one more valid neighbour
                ok = False
                break
            Rm = (Pu[k] - 1) // ratios[k]
        if not ok:
            continue
        # ascending: margin of the post-smoothed u; coarsest solution is exact on all W ghosts
        V = W[nl - 1]
        for k in range(nl - 2, -1, -1):
            V = min(Pu[k], ratios[k] * V) - per * nPost
            if V < 0:
                ok = False
                break
        if ok:
            return W
    raise ValueError("no feasible halo width")


class RankLayout:
    """Element ranges of one rank on every level.
    own[k] = (lo, hi) owned elements, loc[k] = (lo, hi) local domain (owned + ghosts, clipped to
    the global domain), m[k] = DoFs per element, ne[k] = global element count."""

    def __init__(self, n_fine, ratios, block_sizes, world, rank, nPre=3, nPost=3, gs=False):
        tot = int(np.prod(ratios)) if len(ratios) else 1
        if n_fine % (tot * world):
            raise ValueError("fine element count must be divisible by world_size * prod(ratios)")
        self.world, self.rank = world, rank
        self.ratios = tuple(ratios)
        self.m = list(block_sizes)
        self.W = halo_widths(ratios, nPre, nPost, gs)
        self.gs = gs
        self.nPre, self.nPost = nPre, nPost
        self.ne, self.own, self.loc = [], [], []
        f = 1
        per_rank = n_fine // world
        if per_rank < self.W[0]:
            raise ValueError(f"each rank must own at least {self.W[0]} fine elements")
        for k in range(len(ratios) + 1):
            if k > 0:
                f *= ratios[k - 1]
            ne = n_fine // f
            lo, hi = rank * per_rank // f, (rank + 1) * per_rank // f
            self.ne.append(ne)
            self.own.append((lo, hi))
            self.loc.append((max(0, lo - self.W[k]), min(ne, hi + self.W[k])))
            if gs and k < len(ratios) and self.loc[-1][0] % 2:
                raise ValueError("block Gauss-Seidel: a rank's local element range must start on an even element")

    def ghosts(self, k):
        return self.own[k][0] - self.loc[k][0], self.loc[k][1] - self.own[k][1]

    def local_dofs(self, k):
        return (self.loc[k][1] - self.loc[k][0]) * self.m[k]

    def owned_slice(self, k):
        gl, _ = self.ghosts(k)
        return slice(gl * self.m[k], (gl + self.own[k][1] - self.own[k][0]) * self.m[k])

    def owned_ranges(self, k):
        """the owned DoFs of level k as [start, stop) ranges of the local numbering"""
        s = self.owned_slice(k)
        return [(s.start, s.stop)]


class CgRankLayout:
    """Element ranges of one rank for a CG p-chain + DG p=0 hierarchy (BASELINE config 5 shape: CG p_0 >
    p_1 > ... on the same elements, then DG p=0).  Every level has the same elements, so the ghost width is
    the same on all of them: one element per sweep, residual, restriction and prolongation (conservatively
    (nPre + nPost + 4) per smoothed level).  A CG level is numbered vertices-first ON THE RANK'S SUB-MESH
    (src/cg_mesh.jl:37-45,59-65): local vertex v - loc_lo, then the interior nodes element by element; a rank
    owns the vertices [own_lo, own_hi) (the last rank also the final vertex) and the interior nodes of its
    elements.  `exchange_layout(level)` describes the interface slices for aggmg_dist_set_exchange_layout."""

    # elements by which one sweep moves information: point-Jacobi one (a row couples the two elements of a vertex);
    # the element Schwarz smoothers two (src/smoother.jl:1-46: the residual of a node, then the element solves that
    # mix it over the element); the red-black element Gauss-Seidel extension four (two half-sweeps, each a residual
    # and an element solve, the second reading the first one's new values)
    SWEEP_REACH = {"jac": 1, "addSchwarz": 2, "hybridSchwarz": 2, "blockGS": 4}

    def __init__(self, n, ps, world, rank, nPre=3, nPost=3, smoother="jac"):
        if n % world:
            raise ValueError("element count must be divisible by world_size")
        if smoother not in self.SWEEP_REACH:
            raise ValueError(f"unknown CG smoother {smoother!r}")
        self.world, self.rank = world, rank
        self.ps = tuple(ps)
        self.smoother = smoother
        nl = len(ps) + 1
        self.ratios = (1,) * (nl - 1)
        self.m = list(ps) + [1]
        w = len(ps) * (self.SWEEP_REACH[smoother] * (nPre + nPost) + 4)
        per = n // world
        if smoother == "blockGS" and world > 1:
            # the two colours are the parity of the LOCAL element index: every local range starts on an even element
            w += w % 2
            if per % 2:
                raise ValueError("element Gauss-Seidel: every rank must own an even number of elements")
        self.W = [w] * nl
        self.nPre, self.nPost = nPre, nPost
        if per < w + 1:
            raise ValueError(f"each rank must own more than {w} elements")
        lo, hi = rank * per, (rank + 1) * per
        self.ne = [n] * nl
        self.own = [(lo, hi)] * nl
        self.loc = [(max(0, lo - w), min(n, hi + w))] * nl

    def ghosts(self, k):
        return self.own[k][0] - self.loc[k][0], self.loc[k][1] - self.own[k][1]

    def local_dofs(self, k):
        nloc = self.loc[k][1] - self.loc[k][0]
        return nloc * self.m[k] + (1 if k < len(self.ps) else 0)

    def owned_index(self, k):
        """local indices of the owned DoFs of level k, in the order of the global numbering"""
        gl, _ = self.ghosts(k)
        nown = self.own[k][1] - self.own[k][0]
        nloc = self.loc[k][1] - self.loc[k][0]
        if k >= len(self.ps):
            return np.arange(gl, gl + nown)
        q = self.ps[k] - 1
        last = 1 if self.own[k][1] == self.ne[k] else 0
        v = np.arange(gl, gl + nown + last)
        it = (nloc + 1) + np.arange(gl * q, (gl + nown) * q)
        return np.concatenate([v, it])

    def owned_ranges(self, k):
        """the same DoFs as [start, stop) ranges of the local numbering (vertices, then element-interior nodes)"""
        gl, _ = self.ghosts(k)
        nown = self.own[k][1] - self.own[k][0]
        nloc = self.loc[k][1] - self.loc[k][0]
        if k >= len(self.ps):
            return [(gl, gl + nown)]
        q = self.ps[k] - 1
        last = 1 if self.own[k][1] == self.ne[k] else 0
        out = [(gl, gl + nown + last)]
        if q > 0:
            out.append(((nloc + 1) + gl * q, (nloc + 1) + (gl + nown) * q))
        return out

    def global_index(self, k):
        """global numbers of the same DoFs"""
        lo, hi = self.own[k]
        n = self.ne[k]
        if k >= len(self.ps):
            return np.arange(lo, hi)
        q = self.ps[k] - 1
        last = 1 if hi == n else 0
        return np.concatenate([np.arange(lo, hi + last), (n + 1) + np.arange(lo * q, hi * q)])

    def exchange_layout(self, k):
        """-> (count, send, left, right), each a list of (src, dst, len): the first W elements' nodes incl. the
        vertex that closes them go to the left neighbour, the last W elements' nodes to the right one"""
        if k >= len(self.ps):
            return None                                    # contiguous elements: the library's default
        W, q = self.W[k], self.ps[k] - 1
        gl, gr = self.ghosts(k)
        nown = self.own[k][1] - self.own[k][0]
        nloc = self.loc[k][1] - self.loc[k][0]
        I0 = nloc + 1                                      # start of the interior part of the local vector
        # this rank's part: [first: W+1 vertices | first: W q interior | last: W vertices | last: W q interior]
        o_fv, o_fi, o_lv, o_li = 0, W + 1, W + 1 + W * q, 2 * W + 1 + W * q
        count = 2 * W + 1 + 2 * W * q
        send = [(gl, o_fv, W + 1), (I0 + gl * q, o_fi, W * q),
                (gl + nown - W, o_lv, W), (I0 + (gl + nown - W) * q, o_li, W * q)]
        left = [(o_lv, 0, W), (o_li, I0, W * q)] if gl else []                       # left ghosts <- neighbour's last part
        right = [(o_fv, gl + nown, W + 1), (o_fi, I0 + (gl + nown) * q, W * q)] if gr else []
        return count, [s_ for s_ in send if s_[2] > 0], [s_ for s_ in left if s_[2] > 0], [s_ for s_ in right if s_[2] > 0]


    def neighbor_layout(self, k):
        """-> (to_left, to_right, from_left, from_right), each a list of (offset, len) slices of the local level-k
        vector (aggmg_dist_set_neighbor_layout): the same interface as exchange_layout, as neighbour messages --
        vertex slice, then interior slice, in both directions"""
        if k >= len(self.ps):
            return None
        count, send, left, right = self.exchange_layout(k)
        by_dst = {s_[1]: s_ for s_ in send}            # pack offset -> (x offset, pack offset, len)
        W, q = self.W[k], self.ps[k] - 1
        o_fv, o_fi, o_lv, o_li = 0, W + 1, W + 1 + W * q, 2 * W + 1 + W * q
        sl = lambda *offs: [(by_dst[o][0], by_dst[o][2]) for o in offs if o in by_dst]
        return (sl(o_fv, o_fi), sl(o_lv, o_li), [(d, n) for (_, d, n) in left], [(d, n) for (_, d, n) in right])


# ------------------------------------------------------------------------------------------
# communicator (torch.distributed; device tensors for nccl, host staging for gloo)
# ------------------------------------------------------------------------------------------
class Comm:
    def __init__(self, world, rank, staged=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank = world, rank
        self.staged = staged   # True: collectives run on CPU copies (gloo with GPU engines)

    def all_gather(self, out, inp):
        """out (world * len(inp)) <- concatenation over ranks of inp (equal lengths)"""
        if self.world == 1:
            out.copy_(inp)
            return
        if self.staged and inp.is_cuda:
            o = self.torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(o, inp.cpu())
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, inp)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, value):
        if self.world == 1:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64)
        if not self.staged and self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        """sum over ranks of a double, added in rank order on every rank: all ranks hold the same bits (the stopping
        test of a partitioned multigrid() must fall the same way everywhere)"""
        if self.world == 1:
            return float(value)
        on_dev = (not self.staged) and self.dist.get_backend() == "nccl"
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device="cuda" if on_dev else "cpu")
        o = self.torch.empty(self.world, dtype=self.torch.float64, device=t.device)
        self.dist.all_gather_into_tensor(o, t)
        return float(np.sum(np.asarray(o.cpu().tolist())))

    def min_int(self, value, device=None):
        """minimum over ranks of a small integer (agreement on a route before the first collective)"""
        if self.world == 1:
            return int(value)
        on_dev = (not self.staged) and self.dist.get_backend() == "nccl"
        t = self.torch.tensor([int(value)], dtype=self.torch.int32, device=device if on_dev else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    # host-staged forms behind the library's collective callbacks (gloo rehearsals: ranks share a GPU)
    def host_all_gather(self, s):
        """NumPy vector -> concatenation over ranks"""
        o = self.torch.empty(s.size * self.world, dtype=self.torch.float64)
        self.dist.all_gather_into_tensor(o, self.torch.from_numpy(s))
        return o.numpy()

    def host_sendrecv(self, ops, bufs):
        """ops: (peer, is_send, count) in issue order, bufs: one NumPy vector each (filled for sends, to be filled for
        receives); messages of one pair and direction match in order"""
        reqs, tags = [], {}
        for (pr, snd, n), t in zip(ops, bufs):
            tag = tags.get((pr, snd), 0)
            tags[(pr, snd)] = tag + 1
            tt = self.torch.from_numpy(t)
            reqs.append(self.dist.isend(tt, pr, tag=tag) if snd else self.dist.irecv(tt, pr, tag=tag))
        for r_ in reqs:
            r_.wait()


class ThreadGroup:
    """shared state of `world` ranks run as THREADS of one process (ThreadComm)"""

    def __init__(self, world):
        import queue
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.lock = threading.Lock()
        self.boxes = {}
        self._queue = queue.Queue

    def box(self, key):
        with self.lock:
            q = self.boxes.get(key)
            if q is None:
                q = self.boxes[key] = self._queue()
            return q


class ThreadComm(Comm):
    """Comm over threads: `world` ranks inside ONE process, each with its own library context and stream on the same
    GPU, collectives through shared memory and a barrier.  A rehearsal aid -- the partition layouts, the C++ schedule
    of csrc/dist.hip and its callbacks at world sizes the box has neither GPUs nor process slots for (a one-GPU box
    admits six GPU processes; the north-star job has eight ranks).  Host-staged like the gloo tests; nothing is timed
    meaningfully through it."""

    def __init__(self, group, rank):
        import torch
        self.torch, self.dist = torch, None
        self.world, self.rank = group.world, rank
        self.staged = True
        self.g = group

    def _exchange(self, mine):
        g = self.g
        g.slots[self.rank] = mine
        g.barrier.wait()
        allv = list(g.slots)
        g.barrier.wait()
        return allv

    def all_gather(self, out, inp):
        parts = self._exchange(inp.detach().cpu().clone())
        out.copy_(self.torch.cat(parts))

    def barrier(self):
        self.g.barrier.wait()

    def max(self, value):
        return max(self._exchange(float(value)))

    def min_int(self, value, device=None):
        return min(self._exchange(int(value)))

    def sum(self, value):
        return float(np.sum(np.asarray(self._exchange(float(value)))))

    def host_all_gather(self, s):
        return np.concatenate(self._exchange(np.array(s, copy=True)))

    def host_sendrecv(self, ops, bufs):
        tags = {}
        todo = []
        for (pr, snd, n), t in zip(ops, bufs):
            tag = tags.get((pr, snd), 0)
            tags[(pr, snd)] = tag + 1
            if snd:
                self.g.box((self.rank, pr, tag)).put(np.array(t, copy=True))
            else:
                todo.append((pr, tag, t))
        for pr, tag, t in todo:
            t[:] = self.g.box((pr, self.rank, tag)).get(timeout=600)


# ------------------------------------------------------------------------------------------
# the schedule
# ------------------------------------------------------------------------------------------
class DistributedVCycle:
    """multigrid_v_cycle (src/solvers.jl:19-50) on an element-partitioned hierarchy.

    engine interface (vectors are torch tensors on the engine's device, local numbering):
        engine.new(n) -> zero tensor
        engine.down(x0, b, nPre, alpha)          descending half on the local hierarchy
        engine.coarse_rhs() -> tensor            local coarsest right-hand side
        engine.coarse_solve(rhs_global) -> tensor   global coarsest direct solve (replicated)
        engine.set_coarse_solution(t)            local coarsest solution (owned + ghosts)
        engine.up(b, x_out, nPost, alpha)        ascending half
    """

    def __init__(self, engine, layout, comm):
        self.e, self.L, self.c = engine, layout, comm
        L = layout
        self._bufs = {}
        nc = len(L.m) - 1
        self._nc = nc
        self._own_c = (L.own[nc][1] - L.own[nc][0]) * L.m[nc]
        self._rhs_global = engine.new(L.ne[nc] * L.m[nc])
        self.exchanges = 0
        # chunked coarsest solve: every rank reduces the chunks of its own block range, only the
        # chunk-boundary system is gathered (falls back to gather + replicated solve otherwise)
        self.chunked = False
        if L.world > 1 and hasattr(engine, "coarse_plan") and os.environ.get("AGGMG_DIST_COARSE", "chunked") == "chunked":
            q, nq, mblk, nblk = engine.coarse_plan()
            own_blk = L.own[nc][1] - L.own[nc][0]
            if q > 0 and mblk == L.m[nc] and nblk == L.ne[nc] and own_blk % (1 << q) == 0 and own_blk >= (1 << q):
                self.chunked = True
                self._q, self._nq = q, nq
                self._cnt = (own_blk >> q) * mblk
                self._send2 = engine.new(2 * self._cnt)
                self._recv2 = engine.new(2 * self._cnt * L.world)

    def exchange_ghosts(self, x, level=0):
        """Fill the ghost entries of a local level-`level` vector from the neighbours' owned
        boundary elements: all-gather of each rank's first and last W owned elements (the
        interface DoFs)."""
        L = self.L
        if L.world == 1 or L.W[level] == 0:
            return
        m, W = L.m[level], L.W[level]
        gl, gr = L.ghosts(level)
        wm = W * m
        o0 = gl * m
        o1 = o0 + (L.own[level][1] - L.own[level][0]) * m
        send, recv = self._xbuf(level)
        self._copy([(send[:wm], x[o0:o0 + wm]), (send[wm:], x[o1 - wm:o1])])
        self.c.all_gather(recv, send)
        self.exchanges += 1
        self._copy(self._ghost_pairs(x, recv, level))

    def _ghost_pairs(self, x, recv, level):
        """(dst, src) slices that move the neighbours' interface DoFs from an all-gathered
        buffer into the ghost entries of x"""
        L = self.L
        m, W = L.m[level], L.W[level]
        gl, gr = L.ghosts(level)
        wm = W * m
        o0 = gl * m
        o1 = o0 + (L.own[level][1] - L.own[level][0]) * m
        r = L.rank
        pairs = []
        if gl:   # left neighbour's last W elements
            pairs.append((x[:o0], recv[(r - 1) * 2 * wm + wm:(r - 1) * 2 * wm + 2 * wm]))
        if gr:
            pairs.append((x[o1:], recv[(r + 1) * 2 * wm:(r + 1) * 2 * wm + wm]))
        return pairs

    def _copy(self, pairs):
        """dst <- src for a few (dst, src) tensor pairs: ONE launch on the HIP engine
        (aggmg_copy_segments_dev), plain copies otherwise"""
        if not pairs:
            return
        if hasattr(self.e, "copy_segments"):
            self.e.copy_segments(pairs)
        else:
            for d, s_ in pairs:
                d.copy_(s_)

    def _xbuf(self, level):
        if level not in self._bufs:
            wm = self.L.W[level] * self.L.m[level]
            self._bufs[level] = (self.e.new(2 * wm), self.e.new(2 * wm * self.L.world))
        return self._bufs[level]

    # ---- interface exchange of the NEXT cycle's x0 under the fine-level ascent -----------------
    def _begin_exchange(self, x, launch_ends):
        """On the side stream: the fine-level tiles at the two ends of the local domain
        (launch_ends), then pack and all-gather x's interface elements -- while the main stream
        smooths the middle.  Completed by _finish_exchange(x)."""
        L = self.L
        m, W = L.m[0], L.W[0]
        gl, _ = L.ghosts(0)
        wm = W * m
        o0 = gl * m
        o1 = o0 + (L.own[0][1] - L.own[0][0]) * m
        send, recv = self._xbuf(0)
        with self.e.side_stream():            # starts after everything enqueued on the main stream so far
            launch_ends()
            self.e.mark_side_ends()
            send[:wm].copy_(x[o0:o0 + wm])    # plain copies: off the critical path
            send[wm:].copy_(x[o1 - wm:o1])
            self.c.all_gather(recv, send)
        self.exchanges += 1
        self._pending = x

    def _finish_exchange(self, x):
        self.e.wait_side_stream()             # main stream waits for the all-gather
        self._pending = None
        self._copy(self._ghost_pairs(x, self._xbuf(0)[1], 0))

    def vcycle(self, x0, b, x_out, nPre=3, nPost=3, alpha=2.0 / 3.0, x0_ghosts_valid=False, overlap_next=False):
        """x0, b, x_out: local vectors (owned + ghosts).  b must be valid on the whole local domain
        (set once with exchange_ghosts or generated that way); x0's ghosts are refreshed here.  On
        return the owned part of x_out is the V-cycle result; its ghosts are not.

        overlap_next=True (the caller will pass x_out as the next cycle's x0, as the loop of
        multigrid does, src/solvers.jl:124-126): the fine-level ascent produces the interface
        elements of x_out first and their all-gather runs on a second stream under the rest of
        that launch; the next vcycle(x_out, ...) only waits for it and fills the ghosts.  Same
        arithmetic, one collective off the critical path."""
        L = self.L
        if nPre > L.nPre or nPost > L.nPost:
            raise ValueError("halo widths were sized for fewer sweeps")
        pending = getattr(self, "_pending", None)
        if pending is not None:
            if pending is x0:
                self._finish_exchange(x0)
                x0_ghosts_valid = True
            else:                              # stale prefetch: drain it, then exchange as usual
                self.e.wait_side_stream()
                self._pending = None
        if not x0_ghosts_valid:
            self.exchange_ghosts(x0)
        self.e.down(x0, b, nPre, alpha)
        rhs_c = self.e.coarse_rhs()
        nc = self._nc
        mc = L.m[nc]
        gl, _ = L.ghosts(nc)
        own = rhs_c[gl * mc:gl * mc + self._own_c]
        if self.chunked:
            e, cnt, P = self.e, self._cnt, L.world
            blo, bhi = L.own[nc]
            clo = blo >> self._q
            partR, partL = e.coarse_forward(own, blo, bhi)
            self._copy([(self._send2[:cnt], partR[clo * mc:clo * mc + cnt]),
                        (self._send2[cnt:], partL[(clo + 1) * mc:(clo + 1) * mc + cnt])])
            self.c.all_gather(self._recv2, self._send2)
            self.exchanges += 1
            rv = self._recv2.view(P, 2, cnt)
            self._copy([(partR[:P * cnt].view(P, cnt), rv[:, 0]), (partL[mc:mc + P * cnt].view(P, cnt), rv[:, 1])])
            e.coarse_boundary_solve()
            sol_c = e.coarse_solution()
            e.coarse_backward(own, blo, bhi, sol_c[gl * mc:gl * mc + self._own_c])
            self.exchange_ghosts(sol_c, nc)
        else:
            self.c.all_gather(self._rhs_global, own)
            self.exchanges += 1
            sol = self.e.coarse_solve(self._rhs_global)
            lo, hi = L.loc[nc]
            self.e.set_coarse_solution(sol[lo * mc:hi * mc])
        if overlap_next and L.world > 1 and L.W[0] > 0 and getattr(self.e, "can_split_up", lambda n: False)(nPost):
            gl0, _ = L.ghosts(0)
            head = gl0 + L.W[0]                                   # local elements [0, head): left ghosts + first W owned
            tail = gl0 + (L.own[0][1] - L.own[0][0]) - L.W[0]     # [tail, end): last W owned + right ghosts
            self.e.up_split(b, x_out, nPost, alpha, head, tail, 0)       # coarser levels
            self._begin_exchange(x_out, lambda: self.e.up_split(b, x_out, nPost, alpha, head, tail, 1))
            self.e.up_split(b, x_out, nPost, alpha, head, tail, 2)       # the middle of the fine level
            self.e.wait_side_ends()                                     # x_out is whole on the main stream again
        else:
            self.e.up(b, x_out, nPost, alpha)


# ------------------------------------------------------------------------------------------
# the same schedule inside the library (C ABI aggmg_dist_*): no Python between the launches of a cycle
# ------------------------------------------------------------------------------------------
class NativeDistributedVCycle:
    """DistributedVCycle's schedule executed by libaggmg_hip.so (csrc/dist.hip): one C call per cycle,
    collectives issued from C++ on the library's stream.

    collectives = 'rccl'      RCCL inside the library (ncclAllGather); the unique id is broadcast once
                              through the torch.distributed group that is already up
                  'torch'     a callback into torch.distributed (gloo: host-staged, tests on one GPU;
                              nccl: device tensors on the library's stream)
                  'loopback'  device-local copies (rehearsal of one rank's share, measurement only)"""

    def __init__(self, engine, layout, comm, collectives="rccl"):
        from . import _lib
        self.e, self.L, self.c = engine, layout, comm
        ctx = engine.ctx
        self.ctx = ctx
        L = layout
        nl = len(L.m)
        i64 = lambda v: (ctypes.c_int64 * nl)(*[int(x) for x in v])
        i32 = lambda v: (ctypes.c_int32 * nl)(*[int(x) for x in v])
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.aggmg_dist_create(
            ctx.handle, engine.H.handle, engine.Hc.handle, L.world, L.rank, nl,
            i64([o[0] for o in L.own]), i64([o[1] for o in L.own]), i64([o[0] for o in L.loc]), i64([o[1] for o in L.loc]),
            i64(L.ne), i32(L.m), i32(L.W), ctypes.byref(h)))
        self.handle = h
        self.collectives = collectives
        self.rccl_ranks = None
        if hasattr(L, "exchange_layout"):      # levels whose interface is not "the first / last W elements"
            for k in (0, nl - 1):
                lay = L.exchange_layout(k)
                if lay is None:
                    continue
                count, send, left, right = lay
                arr = lambda segs, j: (ctypes.c_int64 * max(len(segs), 1))(*[int(t[j]) for t in segs])
                ctx.check(ctx.lib.aggmg_dist_set_exchange_layout(
                    ctx.handle, h, k, int(count), len(send), arr(send, 0), arr(send, 1), arr(send, 2),
                    len(left), arr(left, 0), arr(left, 1), arr(left, 2), len(right), arr(right, 0), arr(right, 1), arr(right, 2)))
                nbl = L.neighbor_layout(k) if hasattr(L, "neighbor_layout") else None
                if nbl is not None:      # the same interface as neighbour messages (one grouped send/recv, no pack / unpack)
                    a = []
                    for segs in nbl:
                        a += [len(segs), arr(segs, 0), arr(segs, 1)]
                    ctx.check(ctx.lib.aggmg_dist_set_neighbor_layout(ctx.handle, h, k, *a))
        if collectives in ("rccl", "loopback"):
            # nothing of the cycle goes through torch any more: run on the library's own (capturable,
            # non-blocking) stream instead of torch's current one; torch-side initialisation of the
            # vectors is complete after this synchronisation
            engine.torch.cuda.synchronize()
            ctx.reset_stream()
        if collectives == "rccl":
            self._init_rccl()
        elif collectives == "torch":
            self._cb = _lib.ALLGATHER_FN(self._torch_allgather)     # keep the callback objects alive
            ctx.check(ctx.lib.aggmg_dist_set_allgather(ctx.handle, h, ctypes.cast(self._cb, ctypes.c_void_p), None))
            self._cb2 = _lib.SENDRECV_FN(self._torch_sendrecv)
            ctx.check(ctx.lib.aggmg_dist_set_sendrecv(ctx.handle, h, ctypes.cast(self._cb2, ctypes.c_void_p)))
        elif collectives == "loopback":
            ctx.check(ctx.lib.aggmg_dist_set_loopback(ctx.handle, h))
        else:
            raise ValueError("collectives: 'rccl', 'torch' or 'loopback'")

    def _init_rccl(self):
        """rank 0 creates the RCCL unique id, the existing process group broadcasts its 128 bytes"""
        import torch
        from . import _lib
        ctx, c = self.ctx, self.c
        nb = 2 * _lib.RCCL_ID_BYTES        # two ids: a communicator per stream (main, side)
        buf = (ctypes.c_ubyte * nb)()
        failed = None
        if c.rank == 0:
            try:
                for k in range(2):
                    one = (ctypes.c_ubyte * _lib.RCCL_ID_BYTES)()
                    ctx.check(ctx.lib.aggmg_rccl_unique_id(ctx.handle, one, _lib.RCCL_ID_BYTES))
                    buf[k * _lib.RCCL_ID_BYTES:(k + 1) * _lib.RCCL_ID_BYTES] = list(one)
            except Exception as exc:      # still take part in the broadcast below: the other ranks are waiting in it
                failed = exc
        if c.world > 1:
            t = torch.tensor(list(buf) + [0 if failed else 1], dtype=torch.uint8)
            on_dev = (not c.staged) and c.dist.get_backend() == "nccl"
            if on_dev:
                t = t.to(self.e.dev)
            c.dist.broadcast(t, src=0)
            vals = t.cpu().tolist()
            if not vals[-1]:
                raise failed or RuntimeError("rank 0 could not create the RCCL unique id")
            buf = (ctypes.c_ubyte * nb)(*vals[:-1])
        elif failed:
            raise failed
        n = ctypes.c_int(0)
        ctx.check(ctx.lib.aggmg_dist_init_rccl(ctx.handle, self.handle, buf, nb, ctypes.byref(n)))
        self.rccl_ranks = n.value

    def _torch_allgather(self, user, send, recv, count, stream):
        """aggmg_allgather_fn: recv[r * count + i] = rank r's send[i], ordered on `stream`"""
        try:
            import torch
            c = self.c
            if c.staged or c.dist.get_backend() != "nccl":
                ctx = self.ctx
                s = np.empty(int(count))
                ctx.check(ctx.lib.aggmg_memcpy_d2h(ctx.handle, s.ctypes.data, ctypes.c_void_p(send), int(count) * 8))   # syncs the stream
                o = np.ascontiguousarray(c.host_all_gather(s))
                ctx.check(ctx.lib.aggmg_memcpy_h2d(ctx.handle, ctypes.c_void_p(recv), o.ctypes.data, o.size * 8))
            else:
                ext = torch.cuda.ExternalStream(int(stream or 0), device=self.e.dev)
                with torch.cuda.stream(ext):
                    si = torch.as_tensor(_DevView(send, int(count)), device=self.e.dev)
                    ro = torch.as_tensor(_DevView(recv, int(count) * c.world), device=self.e.dev)
                    c.dist.all_gather_into_tensor(ro, si)
            return 0
        except Exception:       # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1

    def _torch_sendrecv(self, user, nops, peer, is_send, ptr, count, stream):
        """aggmg_sendrecv_fn over torch.distributed point-to-point operations (gloo: host-staged; nccl: device
        tensors on the library's stream, one batch)"""
        try:
            import torch
            c = self.c
            ops = [(int(peer[i]), bool(is_send[i]), int(ptr[i]), int(count[i])) for i in range(int(nops))]
            if c.staged or c.dist.get_backend() != "nccl":
                ctx = self.ctx
                bufs = []
                for pr, snd, p_, n in ops:
                    t = np.empty(n)
                    if snd:
                        ctx.check(ctx.lib.aggmg_memcpy_d2h(ctx.handle, t.ctypes.data, ctypes.c_void_p(p_), n * 8))
                    bufs.append(t)
                c.host_sendrecv([(pr, snd, n) for pr, snd, _, n in ops], bufs)
                for (pr, snd, p_, n), t in zip(ops, bufs):
                    if not snd:
                        ctx.check(ctx.lib.aggmg_memcpy_h2d(ctx.handle, ctypes.c_void_p(p_), t.ctypes.data, n * 8))
            else:
                ext = torch.cuda.ExternalStream(int(stream or 0), device=self.e.dev)
                with torch.cuda.stream(ext):
                    p2p = []
                    for pr, snd, p_, n in ops:
                        t = torch.as_tensor(_DevView(p_, n), device=self.e.dev)
                        p2p.append(c.dist.P2POp(c.dist.isend if snd else c.dist.irecv, t, pr))
                    for r_ in c.dist.batch_isend_irecv(p2p):
                        r_.wait()
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def set_coarse_overlap(self, on):
        """the exchange of the coarsest solution's ghost blocks under the middle tiles of the two-level ascent (C ABI
        aggmg_dist_set_coarse_overlap); every rank must choose the same"""
        self.ctx.check(self.ctx.lib.aggmg_dist_set_coarse_overlap(self.ctx.handle, self.handle, int(bool(on))))

    def info(self):
        ex, ch, be = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
        self.ctx.check(self.ctx.lib.aggmg_dist_info(self.ctx.handle, self.handle, ctypes.byref(ex), ctypes.byref(ch), ctypes.byref(be)))
        return dict(exchanges=ex.value, chunked=bool(ch.value), backend=["none", "callback", "rccl", "loopback"][be.value])

    @property
    def exchanges(self):
        return self.info()["exchanges"]

    @property
    def chunked(self):
        return self.info()["chunked"]

    def exchange_ghosts(self, x, level=0):
        self.ctx.check(self.ctx.lib.aggmg_dist_exchange_ghosts_dev(self.ctx.handle, self.handle, _p(x), int(level)))

    def allgather(self, out, inp):
        self.ctx.check(self.ctx.lib.aggmg_dist_allgather_dev(self.ctx.handle, self.handle, _p(inp), _p(out), inp.numel()))

    def graph_info(self):
        r, c, b = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
        self.ctx.check(self.ctx.lib.aggmg_dist_graph_info(self.ctx.handle, self.handle, ctypes.byref(r), ctypes.byref(c), ctypes.byref(b)))
        return dict(replays=r.value, captured=c.value, broken=bool(b.value))

    def vcycle(self, x0, b, x_out, nPre=3, nPost=3, alpha=2.0 / 3.0, x0_ghosts_valid=False, overlap_next=False,
               graph=False):
        from . import _lib
        if nPre > self.L.nPre or nPost > self.L.nPost:
            raise ValueError("halo widths were sized for fewer sweeps")
        flags = ((_lib.DIST_X0_GHOSTS_VALID if x0_ghosts_valid else 0) | (_lib.DIST_OVERLAP_NEXT if overlap_next else 0) |
                 (_lib.DIST_GRAPH if graph else 0))
        self.ctx.check(self.ctx.lib.aggmg_dist_vcycle_dev(self.ctx.handle, self.handle, _p(x0), _p(b), _p(x_out), int(nPre),
                                                           int(nPost), float(alpha), flags))

    def free(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx.lib.aggmg_dist_free(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------
# HIP engine
# ------------------------------------------------------------------------------------------
def main_stream_handle(eng):
    return eng._main_handle


class _DevView:
    """zero-copy torch view of a device buffer owned by libaggmg_hip"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class HipEngine:
    """Local hierarchy (coarsest solve external) + the replicated global coarsest solver, both on
    this rank's GPU; vectors are torch CUDA tensors and the library launches on torch's current
    stream so that torch copies / RCCL collectives and the kernels stay ordered."""

    def __init__(self, H_local, H_coarse, ctx):
        import torch
        self.torch = torch
        self.H, self.Hc, self.ctx = H_local, H_coarse, ctx
        self.dev = torch.device("cuda", ctx.device)
        self._main_handle = torch.cuda.current_stream(self.dev).cuda_stream
        ctx.set_stream(self._main_handle)
        rp, sp_, n = H_local.coarse_buffers()
        self._rhs_c = torch.as_tensor(_DevView(rp, n), device=self.dev)
        self._sol_c = torch.as_tensor(_DevView(sp_, n), device=self.dev)
        self._sol_g = None
        self._partR = self._partL = self._xq = None

    def new(self, n):
        return self.torch.zeros(int(n), dtype=self.torch.float64, device=self.dev)

    def owned_sumsq(self, ranges, b, x=None):
        """sum over the local rows in `ranges` of b^2 (x None) or of (b - A_0 x)^2 with the local fine operator (x: a
        local vector whose ghosts are valid, so that the rows of the owned elements are whole) -- this rank's term of
        ||b||^2 / ||A x - b||^2 (src/solvers.jl:127,131); library launches on the library's stream"""
        c = self.ctx
        v = b
        if x is not None:
            if getattr(self, "_r0", None) is None or self._r0.numel() != x.numel():
                self._r0 = self.new(x.numel())
                self.torch.cuda.synchronize()
            c.check(c.lib.aggmg_residual_dev(c.handle, self.H._ops[0].handle, _p(x), _p(b), _p(self._r0)))
            v = self._r0
        tot, out = 0.0, ctypes.c_double(0.0)
        for lo, hi in ranges:
            ptr = ctypes.c_void_p(v.data_ptr() + 8 * int(lo))
            c.check(c.lib.aggmg_dot_dev(c.handle, ptr, ptr, int(hi - lo), ctypes.byref(out)))
            tot += out.value
        return tot

    def down(self, x0, b, nPre, alpha):
        self.H.vcycle_down_dev(x0, b, nPre, alpha)

    def coarse_rhs(self):
        return self._rhs_c

    def coarse_solve(self, rhs_global):
        if self._sol_g is None:
            self._sol_g = self.new(rhs_global.numel())
            self._zero = self.new(rhs_global.numel())
        # a 1-level hierarchy's V-cycle IS the coarsest direct solve (src/solvers.jl:39)
        self.Hc.vcycle_dev(self._zero, rhs_global, self._sol_g, 0, 0, 1.0)
        return self._sol_g

    def set_coarse_solution(self, t):
        self._sol_c.copy_(t)

    def coarse_solution(self):
        return self._sol_c

    # chunked coarsest solve (aggmg_coarse_* entry points)
    def coarse_plan(self):
        c = self.ctx
        q, nq, m, nb = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int64(0)
        c.check(c.lib.aggmg_coarse_plan(c.handle, self.Hc.handle, ctypes.byref(q), ctypes.byref(nq),
                                        ctypes.byref(m), ctypes.byref(nb)))
        if q.value > 0 and self._partR is None:
            self._partR = self.new((nq.value + 1) * m.value)
            self._partL = self.new((nq.value + 1) * m.value)
            self._xq = self.new((nq.value + 1) * m.value)
        return q.value, nq.value, m.value, nb.value

    def coarse_forward(self, rhs_owned, blk_lo, blk_hi):
        c = self.ctx
        c.check(c.lib.aggmg_coarse_chunk_forward_dev(c.handle, self.Hc.handle, _p(rhs_owned), blk_lo, blk_hi,
                                                     _p(self._partR), _p(self._partL)))
        return self._partR, self._partL

    def coarse_boundary_solve(self):
        c = self.ctx
        c.check(c.lib.aggmg_coarse_boundary_solve_dev(c.handle, self.Hc.handle, _p(self._partR), _p(self._partL),
                                                      _p(self._xq)))

    def coarse_backward(self, rhs_owned, blk_lo, blk_hi, x_owned):
        c = self.ctx
        c.check(c.lib.aggmg_coarse_chunk_backward_dev(c.handle, self.Hc.handle, _p(rhs_owned), blk_lo, blk_hi,
                                                      _p(self._xq), _p(x_owned)))

    def up(self, b, x_out, nPost, alpha):
        self.H.vcycle_up_dev(b, x_out, nPost, alpha)

    # ---- split ascent + second stream (interface exchange under the fine-level launch) --------
    def can_split_up(self, nPost):
        # opt-in: on one GPU with loop-back collectives the second stream's event hand-offs cost
        # ~25 us per cycle, about what one small RCCL all-gather is expected to take
        return bool(self.H.structured_levels()[0]) and os.environ.get("AGGMG_DIST_OVERLAP", "0") == "1"

    def up_split(self, b, x_out, nPost, alpha, head, tail, part):
        # the library launches on whatever stream is current in torch (side_stream() switches both)
        self.H.vcycle_up_split_dev(b, x_out, head, tail, part, nPost, alpha)

    def mark_side_ends(self):
        self._ends_done.record(self._side)

    def wait_side_ends(self):
        self.torch.cuda.current_stream(self.dev).wait_event(self._ends_done)

    def side_stream(self):
        """context: torch ops and collectives issued inside go to a second stream that first waits
        for everything enqueued on the main stream so far; leaving records the completion event"""
        torch = self.torch
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.dev)
            self._side_done = torch.cuda.Event()
            self._ends_done = torch.cuda.Event()
            self._main_mark = torch.cuda.Event()
        eng = self

        class _Ctx:
            def __enter__(self_):
                main = torch.cuda.current_stream(eng.dev)
                eng._main_mark.record(main)
                eng._side.wait_event(eng._main_mark)
                self_.cm = torch.cuda.stream(eng._side)
                self_.cm.__enter__()
                eng.ctx.set_stream(eng._side.cuda_stream)      # library launches follow

            def __exit__(self_, *exc):
                eng._side_done.record(eng._side)
                eng.ctx.set_stream(main_stream_handle(eng))
                return self_.cm.__exit__(*exc)

        return _Ctx()

    def wait_side_stream(self):
        if getattr(self, "_side", None) is not None:
            self.torch.cuda.current_stream(self.dev).wait_event(self._side_done)

    def copy_segments(self, pairs):
        """dst <- src for up to four (dst, src) pairs per launch; 1-D contiguous or 2-D views with
        unit inner stride (aggmg_copy_segments_dev)"""
        c = self.ctx
        for i in range(0, len(pairs), 4):
            grp = pairs[i:i + 4]
            n = len(grp)
            src = (ctypes.c_void_p * n)()
            dst = (ctypes.c_void_p * n)()
            rows, cols = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
            sld, dld = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
            for g, (d, s_) in enumerate(grp):
                if d.shape != s_.shape or d.dim() not in (1, 2) or d.stride(-1) != 1 or s_.stride(-1) != 1:
                    raise ValueError("copy_segments: equal shapes, 1-D or 2-D, unit inner stride")
                src[g], dst[g] = s_.data_ptr(), d.data_ptr()
                if d.dim() == 1:
                    rows[g], cols[g], sld[g], dld[g] = 1, d.numel(), d.numel(), d.numel()
                else:
                    rows[g], cols[g] = d.shape[0], d.shape[1]
                    sld[g], dld[g] = max(s_.stride(0), d.shape[1]), max(d.stride(0), d.shape[1])
            c.check(c.lib.aggmg_copy_segments_dev(c.handle, n, src, dst, rows, cols, sld, dld))


def multigrid(dv, x0, b, maxiter, tol, nPre=3, nPost=3, alpha=2.0 / 3.0, check_every=1):
    """multigrid(H, x0, b, maxiter, tol) (src/solvers.jl:116-139) on an element-partitioned hierarchy: the loop
    `x = multigrid_v_cycle(H, x0, b); x0 = x` (:124-126) with dv's schedule (DistributedVCycle or
    NativeDistributedVCycle), `res[i] = ||A x - b||_2` (:127) as the root of the rank-ordered sum of every rank's owned
    rows, the stopping test `res[i] < tol ||b||` (:131) falling the same way on every rank.  x0, b: local vectors (owned
    + ghosts; b valid on the whole local domain); check_every = 1 is the reference's loop, larger values test every k-th
    cycle (and the last).  -> (x local vector whose owned part is the iterate, cycles, res list).  The reference's `err`
    history needs `A \\ b` of the GLOBAL fine operator (:120) and is not formed here."""
    e, L, c = dv.e, dv.L, dv.c
    own = L.owned_ranges(0)
    if hasattr(e, "torch") and e.torch.cuda.is_available():
        e.torch.cuda.synchronize()     # (x0 / b may have been filled on torch's stream just now)
    nb = np.sqrt(c.sum(e.owned_sumsq(own, b)))
    if maxiter <= 0:           # the reference returns its initial `x = zeros(length(x0))` (:119)
        return e.new(L.local_dofs(0)), 0, []
    bufs = [e.new(L.local_dofs(0)), e.new(L.local_dofs(0))]
    if hasattr(e, "torch") and e.torch.cuda.is_available():
        e.torch.cuda.synchronize()     # torch zero-filled the new vectors on ITS stream; the library may run on its own
    src, res, done, valid = x0, [], 0, False
    for it in range(1, int(maxiter) + 1):
        dst = bufs[it % 2]
        check = it % check_every == 0 or it == maxiter
        # (a checked iterate gets its ghosts for the residual rows below -- the next cycle then starts from valid ghosts;
        # otherwise the interface exchange of the next cycle's x0 rides under the fine-level ascent)
        dv.vcycle(src, b, dst, nPre, nPost, alpha, x0_ghosts_valid=valid, overlap_next=not check)
        src, done, valid = dst, it, False
        if check:
            dv.exchange_ghosts(dst)
            valid = True
            r = float(np.sqrt(c.sum(e.owned_sumsq(own, b, dst))))
            res.append(r)
            if r < tol * nb:
                break
    return src, done, res


def _replicated_coarse_hierarchy(Ac, ctx, world):
    """one-level hierarchy of the GLOBAL coarsest operator (every rank holds it; the chunk elimination follows the
    partition).  A rank runs only its share of the chunks: smaller chunks than on one GPU keep its CUs busy"""
    from . import _lib
    from .api import MeshHierarchy
    if world > 1:
        # (measured on one rank's share of an 8-rank 2^24 job: AGGMG_DIST_COARSE_CHUNK_LOG2 = 9 / 10 / 11 / 12)
        ctx.set_option(_lib.OPT_COARSE_CHUNK_LOG2, int(os.environ.get("AGGMG_DIST_COARSE_CHUNK_LOG2", "10")))
    try:
        return MeshHierarchy(None, [Ac], [], [], ctx=ctx, keep_host=False, coarse_mode=_lib.COARSE_AUTO)
    finally:
        if world > 1:
            ctx.set_option(_lib.OPT_COARSE_CHUNK_LOG2, 12)


def build_local_uniform(n, p, pAgg, ratios, layout, ctx, comm, smoother="blockJac"):
    """Local operators of this rank for the uniform model problem, uploaded through the CSC
    boundary, plus the global coarsest operator assembled from every rank's owned block rows.
    -> (HipEngine, U_local)"""
    import torch
    from . import _lib
    from .api import BlockGaussSeidel, BlockJacobi, DeviceOperator, MeshHierarchy
    from .uniform import UniformDgAggHierarchy, block_tridiag_to_csc, _csc
    Smoother = {"blockJac": BlockJacobi, "blockGS": BlockGaussSeidel}[smoother]
    lo, hi = layout.loc[0]
    U = UniformDgAggHierarchy(n, p=p, pAgg=pAgg, ratios=ratios, elem_range=(lo, hi))
    nl = U.nlevels
    ops, sms = [], []
    for k in range(nl):
        op = DeviceOperator(U.stiffness_csc(k), _lib.OP_STIFFNESS, ctx)
        ops.append(op)
        if k < nl - 1:
            sms.append(Smoother(op, U.descriptor(k).mBlockInds, ctx))
    Ls = [DeviceOperator(U.interpolation_csc(k), _lib.OP_TRANSFER, ctx) for k in range(nl - 1)]
    H = MeshHierarchy([U.descriptor(k) for k in range(nl)], ops, sms, Ls, ctx=ctx, keep_host=False,
                      coarse_mode=_lib.COARSE_EXTERNAL)
    # global coarsest operator: gather the owned block rows (sub, diag, sup) of every rank
    nc = nl - 1
    gl, _ = layout.ghosts(nc)
    no = layout.own[nc][1] - layout.own[nc][0]
    mc = layout.m[nc]
    sub, diag, sup = (np.ascontiguousarray(x[gl:gl + no]) for x in U.levels[nc]['A'])
    # the first / last owned block row of a rank couples to a ghost column that exists locally
    # unless the rank sits at the domain boundary, where the block is zero anyway
    mine = torch.from_numpy(np.stack([sub, diag, sup]).reshape(-1))
    dev = torch.device("cuda", ctx.device) if (comm.world > 1 and not comm.staged and
                                                comm.dist.get_backend() == "nccl") else None
    if dev is not None:
        mine = mine.to(dev)
    allb = torch.empty(mine.numel() * comm.world, dtype=torch.float64, device=mine.device)
    comm.all_gather(allb, mine)
    allb = allb.cpu().numpy().reshape(comm.world, 3, no, mc, mc)
    gsub, gdiag, gsup = (np.concatenate([allb[r, i] for r in range(comm.world)]) for i in range(3))
    colptr, rowval, nzval, N = block_tridiag_to_csc(gsub, gdiag, gsup)
    Ac = DeviceOperator(_csc(colptr, rowval, nzval, (N, N)), _lib.OP_STIFFNESS, ctx)
    Hc = _replicated_coarse_hierarchy(Ac, ctx, comm.world)
    return HipEngine(H, Hc, ctx), U


def build_local_cg(n, ps, layout, ctx, comm, smoother=None):
    """Local operators of this rank for the CG p-chain + DG p=0 model hierarchy (config 5 shape), uploaded
    through the CSC boundary with the sub-mesh's element lists (chain kernels), plus the global coarsest
    (DG p=0) operator assembled from every rank's owned rows.  smoother: cg_smoother's kinds 'jac' (default: the
    layout's), 'addSchwarz', 'hybridSchwarz' (src/smoother.jl:88-139) or the 'blockGS' extension -- the layout must
    have been sized for it (CgRankLayout(..., smoother=)).  -> (HipEngine, U_local)"""
    import torch
    from . import _lib
    from .api import (AdditiveSchwarzSmoother, BlockGaussSeidel, DeviceOperator, HybridSchwarzSmoother, JacobiSmoother,
                      MeshHierarchy)
    from .uniform import UniformCgDgHierarchy, block_tridiag_to_csc, _csc
    smoother = smoother or getattr(layout, "smoother", "jac")
    if CgRankLayout.SWEEP_REACH[smoother] > CgRankLayout.SWEEP_REACH[getattr(layout, "smoother", "jac")]:
        raise ValueError(f"the layout's ghost layers were sized for {layout.smoother!r} sweeps, not {smoother!r}")
    lo, hi = layout.loc[0]
    U = UniformCgDgHierarchy(n, ps=ps, elem_range=(lo, hi))
    nl = U.nlevels
    ops = [DeviceOperator(A, _lib.OP_STIFFNESS, ctx) for A in U.A]
    if smoother == "jac":
        sms = [JacobiSmoother(ops[k], ctx, U.element_nodes(k)) for k in range(nl - 1)]
    else:
        cls = {"addSchwarz": AdditiveSchwarzSmoother, "hybridSchwarz": HybridSchwarzSmoother, "blockGS": BlockGaussSeidel}[smoother]
        sms = [cls(ops[k], U.element_nodes(k), ctx) for k in range(nl - 1)]
    Ls = [DeviceOperator(L, _lib.OP_TRANSFER, ctx) for L in U.L]
    H = MeshHierarchy(None, ops, sms, Ls, ctx=ctx, keep_host=False, coarse_mode=_lib.COARSE_EXTERNAL)
    # global coarsest operator: the owned rows of the local DG p=0 operator of every rank
    nc = nl - 1
    gl, _ = layout.ghosts(nc)
    no = layout.own[nc][1] - layout.own[nc][0]
    sub, diag, sup = (np.ascontiguousarray(x[gl:gl + no]) for x in U.dg0.levels[0]['A'])
    mine = torch.from_numpy(np.stack([sub, diag, sup]).reshape(-1))
    dev = torch.device("cuda", ctx.device) if (comm.world > 1 and not comm.staged and
                                                comm.dist.get_backend() == "nccl") else None
    if dev is not None:
        mine = mine.to(dev)
    allb = torch.empty(mine.numel() * comm.world, dtype=torch.float64, device=mine.device)
    comm.all_gather(allb, mine)
    allb = allb.cpu().numpy().reshape(comm.world, 3, no, 1, 1)
    gsub, gdiag, gsup = (np.concatenate([allb[r, i] for r in range(comm.world)]) for i in range(3))
    colptr, rowval, nzval, N = block_tridiag_to_csc(gsub, gdiag, gsup)
    Ac = DeviceOperator(_csc(colptr, rowval, nzval, (N, N)), _lib.OP_STIFFNESS, ctx)
    Hc = _replicated_coarse_hierarchy(Ac, ctx, comm.world)
    return HipEngine(H, Hc, ctx), U


# ------------------------------------------------------------------------------------------
# bench entry for N > 1 (called by bench.py)
# ------------------------------------------------------------------------------------------
def bench_main(args, rank, world, local_rank, nPre, nPost, alpha, group=None):
    """group: a ThreadGroup -- this rank is one of `world` THREADS of a single process sharing one GPU (bench.py
    --rehearse-threads: the N-rank code path end to end on a one-GPU box; its JSON line says so and is not a measurement)"""
    import torch
    import torch.distributed as dist
    from . import api as mg
    # AGGMG_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    # (host-staged collectives, ranks share devices); the driver's runs use nccl == RCCL.
    backend = "threads" if group is not None else os.environ.get("AGGMG_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    device = local_rank % max(ndev, 1)
    torch.cuda.set_device(device)
    if group is not None:
        comm = ThreadComm(group, rank)
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        comm = Comm(world, rank, staged=(backend != "nccl"))
    ctx = mg.Context(device)
    cfg5 = getattr(args, "dist_config", 4) == 5
    n = 2 ** (args.cg_log2_elems if cfg5 else args.log2_elems)
    ratios = (4, 2, 2)
    t_setup = time.perf_counter()
    cg_sm = getattr(args, "dist_smoother", "jac")
    if cfg5:      # CG p = 4, 2, 1 -> DG p = 0 (BASELINE config 5 shape), chain kernels on every rank's sub-mesh
        if cg_sm == "blockGS":
            alpha = 1.0          # as the single-GPU block_gs_extension leg
        layout = CgRankLayout(n, (4, 2, 1), world, rank, nPre, nPost, smoother=cg_sm)
        engine, U = build_local_cg(n, (4, 2, 1), layout, ctx, comm)
    else:
        layout = RankLayout(n, ratios, [args.p + 1, 2, 2, 2], world, rank, nPre, nPost)
        engine, U = build_local_uniform(n, args.p, 1, ratios, layout, ctx, comm)
    # the cycle runs inside the library (csrc/dist.hip).  Collectives: RCCL from C++ when the process
    # group is nccl; AGGMG_DIST_COLLECTIVES=torch routes them through torch.distributed instead (the
    # gloo rehearsal on a box with fewer GPUs than ranks needs that), =python keeps the Python schedule
    mode = os.environ.get("AGGMG_DIST_COLLECTIVES", "rccl" if backend == "nccl" else "torch")
    rccl_note = None
    if mode == "python":
        dv = DistributedVCycle(engine, layout, comm)
    else:
        dv, err = None, None
        try:
            dv = NativeDistributedVCycle(engine, layout, comm, collectives=mode)
        except Exception as exc:         # RCCL not loadable / communicator failed
            if mode != "rccl":
                raise
            err = exc
        if mode == "rccl":
            # every rank has to take the same route: agree on the outcome before the first collective, and fall
            # back together to the same schedule with its all-gathers issued through torch.distributed
            if comm.min_int(0 if dv is None else 1, engine.dev) == 0:
                rccl_note = f"in-library RCCL unavailable on at least one rank ({err!r}); collectives through torch.distributed"
                dv = NativeDistributedVCycle(engine, layout, comm, collectives="torch")
    b = torch.from_numpy(U.rhs()).to(engine.dev)        # generated on the whole local domain
    xa = engine.new(layout.local_dofs(0))
    xb = engine.new(layout.local_dofs(0))
    bytes_model = U.algorithmic_bytes(nPre, nPost)
    del U
    t_setup = time.perf_counter() - t_setup
    N = (4 * n + 1) if cfg5 else n * (args.p + 1)

    # Self-check before anything is timed (untimed, two cycles): the in-library RCCL path -- grouped ncclSend / ncclRecv
    # between the vectors, the in-place ncclAllGather, a communicator per stream -- has only ever run with one rank on
    # the build boxes.  The same schedule with its exchanges routed through torch.distributed (its own NCCL calls, the
    # path every torch job uses) must give the same bits; if it does not, the run goes on with the torch-routed
    # collectives and says so, rather than reporting a rate for wrong numbers.
    selfcheck = None
    if (isinstance(dv, NativeDistributedVCycle) and dv.collectives == "rccl" and
            (world > 1 or os.environ.get("AGGMG_DIST_SELFCHECK") == "force") and os.environ.get("AGGMG_DIST_SELFCHECK") != "0"):
        try:
            dv_t = NativeDistributedVCycle(engine, layout, comm, collectives="torch")
            outs = []
            for drv in (dv, dv_t):
                p_, q_ = engine.new(layout.local_dofs(0)), engine.new(layout.local_dofs(0))
                for _ in range(2):
                    drv.vcycle(p_, b, q_, nPre, nPost, alpha, overlap_next=True)
                    p_, q_ = q_, p_
                torch.cuda.synchronize()
                outs.append(p_[layout.owned_slice(0)] if hasattr(layout, "owned_slice") else p_[torch.as_tensor(layout.owned_index(0), device=p_.device)])
            if comm.min_int(1 if torch.equal(outs[0], outs[1]) else 0, engine.dev) == 1:
                selfcheck = "owned values after two cycles bitwise equal to the torch.distributed-routed schedule"
                dv_t.free()
            else:
                selfcheck = "MISMATCH against the torch.distributed-routed schedule: timed run uses torch.distributed collectives"
                rccl_note = selfcheck
                dv.free()
                dv = dv_t
        except Exception as exc:      # the check must never take the run down
            selfcheck = f"not completed ({exc!r})"

    # Schedule choice measured, not assumed (untimed warm-up): the coarse ghost exchange under the two-level ascent's middle
    # tiles trades an extra launch and two stream joins for the latency of one neighbour exchange -- 9 us lost against 5 us
    # hidden with loop-back stand-ins, a gain where a real exchange takes 15 - 25 us.  Both are timed, the max over ranks of
    # each compared, and every rank keeps the same one.
    coarse_overlap = None
    if isinstance(dv, NativeDistributedVCycle) and world > 1 and dv.chunked and os.environ.get("AGGMG_DIST_COARSE_OVERLAP") is None:
        tune = {}
        p_, q_ = engine.new(layout.local_dofs(0)), engine.new(layout.local_dofs(0))
        for on in (0, 1):
            dv.set_coarse_overlap(on)
            for _ in range(3):
                dv.vcycle(p_, b, q_, nPre, nPost, alpha, overlap_next=True)
                p_, q_ = q_, p_
            torch.cuda.synchronize()
            comm.barrier()
            t1 = time.perf_counter()
            for _ in range(20):
                dv.vcycle(p_, b, q_, nPre, nPost, alpha, overlap_next=True)
                p_, q_ = q_, p_
            torch.cuda.synchronize()
            tune[on] = comm.max(time.perf_counter() - t1) / 20
        keep = 1 if tune[1] < tune[0] else 0
        dv.set_coarse_overlap(keep)
        coarse_overlap = {"chosen": bool(keep), "ms_per_cycle_off": 1e3 * tune[0], "ms_per_cycle_on": 1e3 * tune[1]}
        del p_, q_

    src, dst = xa, xb
    # every cycle's output is the next cycle's x0 (the loop of multigrid, src/solvers.jl:124-126):
    # its interface exchange is issued under the fine-level ascent.  AGGMG_DIST_GRAPH=1 lets the library
    # replay the cycle as a hipGraph (built in untimed pre-warm cycles: first call eager, second captured,
    # per buffer order); off by default: on one rank's share of an 8-rank 2^24 job the replayed cycle
    # measured 0.452 ms against 0.430 ms launch by launch (the cycle is GPU-bound, issue time 0.07 ms)
    use_graph = isinstance(dv, NativeDistributedVCycle) and os.environ.get("AGGMG_DIST_GRAPH", "0") == "1"
    kw = dict(overlap_next=True)
    if use_graph:
        kw["graph"] = True
        for _ in range(8):
            dv.vcycle(src, b, dst, nPre, nPost, alpha, **kw)
            src, dst = dst, src
    for _ in range(args.warmup):
        dv.vcycle(src, b, dst, nPre, nPost, alpha, **kw)
        src, dst = dst, src
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    ex0 = dv.exchanges
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dv.vcycle(src, b, dst, nPre, nPost, alpha, **kw)
        src, dst = dst, src
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ex_per_cycle = (dv.exchanges - ex0) // max(args.steps, 1)
    # per-kernel times: an untimed pass with events around the dominant kernel (events cannot ride in a graph)
    ctx.profile_enable(2)
    for _ in range(args.steps):
        dv.vcycle(src, b, dst, nPre, nPost, alpha, overlap_next=True)
        src, dst = dst, src
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    prof = ctx.profile_collect()
    dt = comm.max(dt)
    # the partitioned multigrid() loop to ||A x - b|| < 1e-8 ||b|| from a zero guess (outside the timed region; residual
    # norms summed over the ranks' owned rows; AGGMG_BENCH_DIST_OUTER=0 skips it)
    outer = None
    if os.environ.get("AGGMG_BENCH_DIST_OUTER", "1") != "0":
        t2 = time.perf_counter()
        _, ncyc, hist = multigrid(dv, engine.new(layout.local_dofs(0)), b, 200, 1e-8, nPre, nPost, alpha, check_every=4)
        torch.cuda.synchronize()
        outer = {"multigrid": {"cycles": ncyc, "ms": 1e3 * comm.max(time.perf_counter() - t2), "final_residual": hist[-1],
                               "check_every": 4}}
    if rank == 0:
        cand = {k: v for k, v in prof.items() if k[0] in ("fused_down", "fused_up") and k[1] < len(bytes_model)}
        (dkind, dlevel), (dms, dcnt) = max(cand.items(), key=lambda kv: kv[1][0])
        lm = bytes_model[dlevel]
        per_launch = {"fused_down": nPre * lm['sweep'] + lm['residual'] + lm['restrict'],
                      "fused_up": nPost * lm['sweep'] + lm['prolong']}[dkind]
        kd = {"fused_down": "down", "fused_up": "up"}[dkind]
        comp = sum(engine.H.launch_bytes(dlevel, kd))       # compulsory bytes of rank 0's launch: every array once, as stored
        achieved = comp / (dms / dcnt * 1e-3) / 1e9
        out = {
            "metric": "fine_level_dof_updates_per_s_per_vcycle",
            "value": N * (nPre + nPost) * args.steps / dt,
            "unit": "DoF-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"config 5 shape: CG n=2^{args.cg_log2_elems} p=4 -> 2 -> 1 -> DG p=0, "
                                    + {"jac": "point-Jacobi", "addSchwarz": "additive element Schwarz", "hybridSchwarz": "hybrid element Schwarz",
                                       "blockGS": "red-black element Gauss-Seidel (EXTENSION, alpha = 1)"}[cg_sm] + ", V(3,3), "
                                    f"partitioned by contiguous element range over {world} GPUs" if cfg5 else
                                    f"config 4: config 3 hierarchy (DG p={args.p} n=2^{args.log2_elems} -> AggDG 4:1 -> "
                                    f"2:1 -> 2:1, V(3,3)) partitioned by contiguous element range over {world} GPUs"),
                       "fine_dofs": N, "nPre": nPre, "nPost": nPost,
                       "backend": backend,
                       "collectives": (getattr(dv, "collectives", "python schedule") if rccl_note is None else rccl_note),
                       "rccl_ranks": getattr(dv, "rccl_ranks", None),
                       "rccl_selfcheck": selfcheck,
                       "hipgraph": (dv.graph_info() if use_graph else None),
                       "coarse_ghost_exchange_overlapped": coarse_overlap,
                       "parallelism": f"element-range x{world}, deep halos W={layout.W}, "
                                      f"{ex_per_cycle} exchanges per cycle (interface elements as grouped neighbour send/recv "
                                      f"straight between the vectors"
                                      + ("" if os.environ.get("AGGMG_DIST_P2P", "1") != "0" else " -- AGGMG_DIST_P2P=0: pack, all-gather, unpack")
                                      + f"; x0 interface exchange issued under the fine-level ascent), "
                                      + ("coarsest solve: chunk elimination on each rank's own blocks, boundary system replicated"
                                         if dv.chunked else "coarsest solve gathered and replicated")},
            "roofline": {"bound": "hbm", "kernel": (f"cgt_fused_kernel<4> {dkind} level {dlevel + 1} (rank 0)" if cfg5 else
                                                    f"btd_fused_kernel<{args.p + 1},cmp> {dkind} level {dlevel + 1} (rank 0)"),
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "frac_basis": "compulsory bytes (arrays of the launch, each once) / HIP-event time / peak",
                         "traffic": None, "compulsory_bytes_per_launch": comp,
                         "frac_survey_model": per_launch / (dms / dcnt * 1e-3) / 1e9 / 8000.0, "survey_model_bytes_per_launch": per_launch,
                         "ms_per_launch": dms / dcnt, "launches_timed": dcnt},
            "kernels": {f"{k}_L{l}": {"ms_per_launch": v[0] / v[1], "launches": v[1]} for (k, l), v in sorted(prof.items())},
            "outer_solvers_to_1e-8": outer,
            "setup_s": t_setup,
        }
        if group is not None:
            out["rehearsal"] = (f"{world} ranks as threads of ONE process on one GPU (host-staged collectives through shared "
                                "memory): exercises the N-rank code path, NOT a multi-GPU measurement")
        print(json.dumps(out), flush=True)
    if group is None:
        dist.destroy_process_group()
