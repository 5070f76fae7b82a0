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
