"""N > 1 path on CPU: the element-partition / deep-halo schedule of
agglomerationmultigrid1d_amd.distributed run by world_size-2 (and 4) `gloo` process groups with a
NumPy engine (oracle arithmetic) in place of the GPU kernels.  The owned part of every rank's
result must equal the single-domain oracle V-cycle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, p, ratios, nPre, nPost, shrink, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import aggmg_oracle as o
    from dist_helpers import LocalRef, NumpyEngine
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ms = [p + 1] + [2] * len(ratios)
        layout = D.RankLayout(n, ratios, ms, world, rank, nPre, nPost)
        if shrink:   # deliberately too-thin halos: the result must then differ
            layout.W = [max(w - shrink * int(np.prod(ratios[k:])), 0) for k, w in enumerate(layout.W)]
            layout.loc = [(max(0, lo - layout.W[k]), min(layout.ne[k], hi + layout.W[k]))
                          for k, (lo, hi) in enumerate(layout.own)]
        U = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios, elem_range=layout.loc[0])
        Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
        eng = NumpyEngine(o, LocalRef(o, U), Ug.stiffness_csc(Ug.nlevels - 1))
        comm = D.Comm(world, rank)
        dv = D.DistributedVCycle(eng, layout, comm)
        b = torch.from_numpy(U.rhs().copy())
        # a non-trivial x0 whose ghosts are wrong on purpose: exchange_ghosts must repair them
        x0g = o.splitmix_normal(n * (p + 1), 7)
        lo, hi = layout.loc[0]
        x0 = torch.from_numpy(x0g[lo * (p + 1):hi * (p + 1)].copy())
        sl = layout.owned_slice(0)
        bad = x0.clone()
        bad[:sl.start] = 123.0
        bad[sl.stop:] = -321.0
        out = eng.new(layout.local_dofs(0))
        dv.vcycle(bad, b, out, nPre, nPost, 2.0 / 3.0)
        assert torch.equal(bad, x0), "ghost exchange did not restore the neighbours' values"
        # single-domain reference
        Hg = LocalRef(o, Ug)
        xr = o.multigrid_v_cycle(Hg, x0g, Ug.rhs(), nPre=nPre, nPost=nPost, alpha=2.0 / 3.0)
        own_lo, own_hi = layout.own[0]
        ref = xr[own_lo * (p + 1):own_hi * (p + 1)]
        got = out.numpy()[sl]
        err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
        q.put((rank, err, dv.exchanges, list(layout.W)))
    finally:
        dist.destroy_process_group()


def run(world, n, p, ratios, nPre, nPost, shrink=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, p, ratios, nPre, nPost, shrink, q))
             for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    return sorted(q.get() for _ in range(world))


def test_halo_widths():
    from agglomerationmultigrid1d_amd.distributed import RankLayout, halo_widths
    W = halo_widths((4, 2, 2), 3, 3)
    assert W == [80, 20, 10, 5]
    assert halo_widths((4, 2, 2), 1, 1)[0] < W[0] and halo_widths((2,), 3, 3) == [6, 3]
    L = RankLayout(2**12, (4, 2, 2), [4, 2, 2, 2], 4, 1)
    assert L.own[0] == (1024, 2048) and L.loc[0] == (1024 - 80, 2048 + 80) and L.loc[3] == (64 - 5, 128 + 5)
    assert RankLayout(2**12, (4, 2, 2), [4, 2, 2, 2], 4, 0).loc[0] == (0, 1024 + 80)
    with pytest.raises(ValueError):
        RankLayout(2**12, (4, 2, 2), [4, 2, 2, 2], 3, 0)     # not divisible
    with pytest.raises(ValueError):
        RankLayout(256, (4, 2, 2), [4, 2, 2, 2], 4, 0)       # owned range thinner than the halo


def test_two_ranks_match_single_domain():
    res = run(2, 512, 3, (4, 2, 2), 3, 3)
    for rank, err, nex, W in res:
        assert err < 1e-14, (rank, err)     # same per-row arithmetic: equal to the last bits
        assert nex == 2                      # one interface all-gather + the coarsest gather
        assert W == [80, 20, 10, 5]


def test_four_ranks_other_shape():
    for rank, err, nex, W in run(4, 512, 2, (2, 2), 2, 1):
        assert err < 1e-14, (rank, err)


def test_thin_halo_is_detected():
    """with ghost layers one coarsest element thinner than derived the owned result goes wrong:
    the widths of halo_widths() are necessary, not just sufficient"""
    errs = [e for _, e, _, _ in run(2, 512, 3, (4, 2, 2), 3, 3, shrink=1)]
    assert max(errs) > 1e-14     # no longer equal to the last bits (errors enter damped, but they enter)


def _worker_overlap(rank, world, port, n, p, ratios, q):
    """two cycles, the second one's x0 interface exchange prefetched by the first
    (overlap_next=True): schedule bookkeeping of DistributedVCycle on the CPU engine"""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import aggmg_oracle as o
    from dist_helpers import LocalRef, NumpyEngine
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ms = [p + 1] + [2] * len(ratios)
        layout = D.RankLayout(n, ratios, ms, world, rank, 3, 3)
        U = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios, elem_range=layout.loc[0])
        Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
        eng = NumpyEngine(o, LocalRef(o, U), Ug.stiffness_csc(Ug.nlevels - 1))
        dv = D.DistributedVCycle(eng, layout, D.Comm(world, rank))
        b = torch.from_numpy(U.rhs().copy())
        x = eng.new(layout.local_dofs(0))
        y = eng.new(layout.local_dofs(0))
        z = eng.new(layout.local_dofs(0))
        dv.vcycle(x, b, y, overlap_next=True)
        assert dv._pending is y
        dv.vcycle(y, b, z, overlap_next=True)          # consumes the prefetched exchange
        n_after_two = dv.exchanges
        dv.vcycle(x, b, y)                             # a different x0: the stale prefetch is drained
        assert dv._pending is None
        Hg = LocalRef(o, Ug)
        xr = o.multigrid_v_cycle(Hg, np.zeros(n * (p + 1)), Ug.rhs())
        xr2 = o.multigrid_v_cycle(Hg, xr, Ug.rhs())
        lo, hi = layout.own[0]
        sl = layout.owned_slice(0)
        e2 = float(np.max(np.abs(z.numpy()[sl] - xr2[lo * (p + 1):hi * (p + 1)])) / np.max(np.abs(xr2)))
        e1 = float(np.max(np.abs(y.numpy()[sl] - xr[lo * (p + 1):hi * (p + 1)])) / np.max(np.abs(xr)))
        q.put((rank, e1, e2, n_after_two, dv.exchanges))
    finally:
        dist.destroy_process_group()


def test_overlapped_interface_exchange_bookkeeping():
    world, n, p, ratios = 2, 256, 3, (4, 2, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_overlap, args=(r, world, port, n, p, ratios, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for rank, e1, e2, n2, n3 in sorted(q.get() for _ in range(world)):
        assert e1 < 1e-12 and e2 < 1e-12, (rank, e1, e2)
        # replicated coarsest solve at this size: (x0 + rhs gather + prefetch) + (rhs gather + prefetch)
        assert n2 == 5 and n3 == 7


def _thread_ranks(world, fn):
    """run fn(rank, comm) on `world` ThreadComm ranks (threads of this process); -> list of results by rank"""
    import threading
    from agglomerationmultigrid1d_amd import distributed as D
    g = D.ThreadGroup(world)
    out, errs = [None] * world, []

    def one(r):
        try:
            out[r] = fn(r, D.ThreadComm(g, r))
        except BaseException as exc:
            errs.append((r, exc))
            g.barrier.abort()

    ts = [threading.Thread(target=one, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, errs
    return out


@pytest.mark.parametrize("world,n,p,ratios", [(8, 2048, 3, (4, 2, 2)), (8, 1024, 2, (2, 2)), (6, 6 * 256, 3, (4, 2, 2))])
def test_eight_ranks_as_threads_match_single_domain(world, n, p, ratios):
    """the partition at the north-star world size (8 ranks: first / last rank one-sided, six interior ranks with two
    neighbours) with the Python schedule and the oracle's arithmetic, ranks as threads over ThreadComm: owned values equal
    the single-domain oracle V-cycle to the last bits"""
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import aggmg_oracle as o
    from dist_helpers import LocalRef, NumpyEngine
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
    x0g = o.splitmix_normal(n * (p + 1), 7)
    xr = o.multigrid_v_cycle(LocalRef(o, Ug), x0g, Ug.rhs(), nPre=3, nPost=3, alpha=2.0 / 3.0)
    Ac = Ug.stiffness_csc(Ug.nlevels - 1)

    def rank_fn(rank, comm):
        layout = D.RankLayout(n, ratios, [p + 1] + [2] * len(ratios), world, rank, 3, 3)
        U = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios, elem_range=layout.loc[0])
        dv = D.DistributedVCycle(NumpyEngine(o, LocalRef(o, U), Ac), layout, comm)
        lo, hi = layout.loc[0]
        x0 = torch.from_numpy(x0g[lo * (p + 1):hi * (p + 1)].copy())
        out = torch.zeros(layout.local_dofs(0), dtype=torch.float64)
        dv.vcycle(x0, torch.from_numpy(U.rhs().copy()), out, 3, 3, 2.0 / 3.0)
        a, b_ = layout.own[0]
        ref = xr[a * (p + 1):b_ * (p + 1)]
        return float(np.max(np.abs(out.numpy()[layout.owned_slice(0)] - ref)) / np.max(np.abs(ref)))

    errs = _thread_ranks(world, rank_fn)
    assert max(errs) < 1e-14, errs


def test_thread_comm_collectives():
    """ThreadComm's own pieces: all_gather, max / min, matched neighbour messages in issue order"""
    def rank_fn(rank, comm):
        w = comm.world
        out = torch.empty(3 * w, dtype=torch.float64)
        comm.all_gather(out, torch.full((3,), float(rank), dtype=torch.float64))
        assert out.tolist() == [float(r) for r in range(w) for _ in range(3)]
        assert comm.max(rank) == w - 1 and comm.min_int(rank + 5) == 5
        assert np.array_equal(comm.host_all_gather(np.array([rank, -rank], dtype=float)),
                              np.array([v for r in range(w) for v in (r, -r)], dtype=float))
        ops, bufs = [], []
        for peer in (rank - 1, rank + 1):
            if 0 <= peer < w:
                for k in range(2):      # two messages per neighbour and direction: matched in order
                    ops.append((peer, True, 2)); bufs.append(np.array([rank, k], dtype=float))
                    ops.append((peer, False, 2)); bufs.append(np.empty(2))
        comm.host_sendrecv(ops, bufs)
        seen = {}
        for (peer, snd, _), t in zip(ops, bufs):
            if not snd:
                k = seen.get(peer, 0)
                seen[peer] = k + 1
                assert t.tolist() == [float(peer), float(k)]
        comm.barrier()
        return True

    assert all(_thread_ranks(8, rank_fn))


def _mg_worker(rank, world, port, n, p, ratios, maxiter, tol, check_every, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import aggmg_oracle as o
    from dist_helpers import LocalRef, NumpyEngine
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        layout = D.RankLayout(n, ratios, [p + 1] + [2] * len(ratios), world, rank, 3, 3)
        U = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios, elem_range=layout.loc[0])
        Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
        dv = D.DistributedVCycle(NumpyEngine(o, LocalRef(o, U), Ug.stiffness_csc(Ug.nlevels - 1)), layout, D.Comm(world, rank))
        x0 = torch.zeros(layout.local_dofs(0), dtype=torch.float64)
        x, it, res = D.multigrid(dv, x0, torch.from_numpy(U.rhs().copy()), maxiter, tol, check_every=check_every)
        q.put((rank, it, res, x.numpy()[layout.owned_slice(0)].copy(), layout.own[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,check_every,tol", [(2, 1, 2e-4), (2, 3, 2e-4), (3, 1, 1e-30)])
def test_partitioned_multigrid_loop_matches_single_domain(world, check_every, tol):
    """distributed.multigrid -- the loop of multigrid (src/solvers.jl:116-139) over the partitioned V-cycle, residual
    norms summed over the ranks' owned rows, gloo ranks as processes -- against the oracle's multigrid on the whole
    domain: the same cycle count on every rank, the same residual history (to the order of the sums), the same iterate"""
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import aggmg_oracle as o
    from dist_helpers import LocalRef
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    n, p, ratios, maxiter = 96 * world, 3, (4, 2), 7 if tol < 1e-20 else 20   # (2e-4: met after 8 cycles)
    Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
    bg = Ug.rhs()
    Hg = LocalRef(o, Ug)
    _, _, reso, _ = o.multigrid(Hg, np.zeros(len(bg)), bg, maxiter, 1e-30)          # the whole history
    checked = [i for i in range(1, maxiter + 1) if i % check_every == 0 or i == maxiter]
    # the loop stops at the first CHECKED cycle whose residual meets the tolerance
    stop = next((i for i in checked if reso[i - 1] < tol * np.linalg.norm(bg)), maxiter)
    want = [reso[i - 1] for i in checked if i <= stop]
    xo = o.multigrid(Hg, np.zeros(len(bg)), bg, stop, 1e-30)[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_mg_worker, args=(r, world, port, n, p, ratios, maxiter, tol, check_every, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    out = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for pr in procs:
        pr.join(300)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    assert stop < maxiter or tol < 1e-20
    for rank, it, res, xown, (lo, hi) in out:
        assert it == stop and res == out[0][2], (rank, it, stop)          # every rank: the same history, bit for bit
        assert np.allclose(res, want, rtol=1e-9, atol=1e-13 * np.linalg.norm(bg)), (rank, res, want)
        assert np.max(np.abs(xown - xo[lo * (p + 1):hi * (p + 1)])) <= 1e-13 * np.max(np.abs(xo))


def test_eight_thread_ranks_partitioned_multigrid_loop():
    """the same loop on eight ranks (threads, ThreadComm.sum in rank order)"""
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import aggmg_oracle as o
    from dist_helpers import LocalRef, NumpyEngine
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    world, n, p, ratios = 8, 1024, 3, (4, 2, 2)
    Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
    bg = Ug.rhs()
    xo, ito, reso, _ = o.multigrid(LocalRef(o, Ug), np.zeros(len(bg)), bg, 20, 2e-4)
    Ac = Ug.stiffness_csc(Ug.nlevels - 1)

    def rank_fn(rank, comm):
        layout = D.RankLayout(n, ratios, [p + 1] + [2] * len(ratios), world, rank, 3, 3)
        U = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios, elem_range=layout.loc[0])
        dv = D.DistributedVCycle(NumpyEngine(o, LocalRef(o, U), Ac), layout, comm)
        x, it, res = D.multigrid(dv, torch.zeros(layout.local_dofs(0), dtype=torch.float64), torch.from_numpy(U.rhs().copy()), 20, 2e-4)
        lo, hi = layout.own[0]
        return it, res, float(np.max(np.abs(x.numpy()[layout.owned_slice(0)] - xo[lo * (p + 1):hi * (p + 1)])))

    out = _thread_ranks(world, rank_fn)
    for it, res, err in out:
        assert it == ito and res == out[0][1]
        assert np.allclose(res, reso, rtol=1e-9, atol=1e-13 * np.linalg.norm(bg))
        assert err <= 1e-13 * np.max(np.abs(xo))
