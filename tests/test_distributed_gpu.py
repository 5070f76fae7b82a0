"""N > 1 path with the real HIP engine: two ranks sharing the one GPU of the test box, `gloo`
process group with host-staged collectives (RCCL needs one device per rank, so the nccl backend
itself can only run on the multi-GPU node).  Owned results must equal the single-GPU V-cycle of
the same library to the last bits and the oracle within tolerance."""
import os
import sys

import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q, overlap=False, native=False, smoother="blockJac", p2p=True):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import torch
    import torch.distributed as dist
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if overlap:
        os.environ["AGGMG_DIST_OVERLAP"] = "1"
    os.environ["AGGMG_DIST_P2P"] = "1" if p2p else "0"    # neighbour messages (default) or pack -> all-gather -> unpack
    try:
        ratios, p = (4, 2, 2), 3
        ctx = mg.Context(0)
        comm = D.Comm(world, rank, staged=True)
        layout = D.RankLayout(n, ratios, [p + 1, 2, 2, 2], world, rank, gs=(smoother == "blockGS"))
        engine, U = D.build_local_uniform(n, p, 1, ratios, layout, ctx, comm, smoother=smoother)
        assert all(engine.H.structured_levels()) and engine.Hc.coarse_info()['on_device']
        if os.environ.get("AGGMG_TEST_EXPECT_PARALLEL_TAIL"):   # (set by the test that asks for it)
            info = engine.Hc.coarse_info()
            assert info['tail'] == 'parallel cyclic reduction' and 512 < info['tail_blocks'] <= 1024, info
        if native:    # the schedule inside libaggmg_hip.so; its all-gathers call back into torch.distributed (gloo)
            dv = D.NativeDistributedVCycle(engine, layout, comm, collectives="torch")
        else:
            dv = D.DistributedVCycle(engine, layout, comm)
        b = torch.from_numpy(U.rhs()).to(engine.dev)
        x = engine.new(layout.local_dofs(0))
        y = engine.new(layout.local_dofs(0))
        for _ in range(3):                       # three cycles: iterates with non-trivial ghosts
            dv.vcycle(x, b, y, overlap_next=overlap)
            x, y = y, x
        torch.cuda.synchronize()
        got = x.cpu().numpy()[layout.owned_slice(0)]
        # single-GPU run of the same library on the global hierarchy
        Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
        ctx2 = mg.Context(0)
        Hg = build_device_hierarchy(Ug, ctx2, smoother=smoother)
        bg = ctx2.to_device(Ug.rhs())
        xa, xb = ctx2.to_device(np.zeros(len(Ug.rhs()))), ctx2.alloc(len(Ug.rhs()))
        for _ in range(3):
            Hg.vcycle_dev(xa, bg, xb)
            xa, xb = xb, xa
        ref = xa.download()
        lo, hi = layout.own[0]
        ref_own = ref[lo * (p + 1):hi * (p + 1)]
        err = float(np.max(np.abs(got - ref_own)))
        q.put((rank, err, float(np.max(np.abs(ref_own))), dv.exchanges, dv.chunked))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_match_single_gpu():
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 2048, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for rank, err, scale, nex, chunked in sorted(q.get() for _ in range(2)):
        assert err == 0.0, (rank, err, scale)    # bitwise: same per-row arithmetic on every rank
        assert not chunked and nex == 6          # small coarsest level: gather + replicated solve, 2 per cycle


def test_two_ranks_chunked_coarse_solve():
    """coarsest level large enough (2^13 rows) for the chunked cyclic reduction: every rank
    eliminates its own chunks, only the chunk-boundary system and the coarse ghosts are exchanged"""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 2**16, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(900)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for rank, err, scale, nex, chunked in sorted(q.get() for _ in range(2)):
        assert chunked and nex == 9              # x0 ghosts, boundary system, coarse ghosts
        assert err == 0.0, (rank, err, scale)


def test_four_ranks_chunked_coarse_solve():
    """four ranks on the one GPU (the box allows six GPU processes): interior ranks have ghosts on
    both sides and two neighbours in every exchange"""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, 2**16, q)) for r in range(4)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(900)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for rank, err, scale, nex, chunked in sorted(q.get() for _ in range(4)):
        assert chunked and nex == 9
        assert err == 0.0, (rank, err, scale)


def test_two_ranks_boundary_system_of_1024_blocks(monkeypatch):
    """chunks of 4 blocks: the gathered chunk-boundary system has 1024 blocks and is read chunk-interleaved as the
    all-gather left it -- the parallel cyclic reduction of the tail with one ordinary reduction level around it
    (cr_pcr_tail_kernel<M, true>) on strided input.  The single-GPU reference run plans its own chunks and its own
    tail, so the iterates agree to the accuracy of the coarsest solves, not bit for bit."""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    monkeypatch.setenv("AGGMG_DIST_COARSE_CHUNK_LOG2", "2")
    monkeypatch.setenv("AGGMG_TEST_EXPECT_PARALLEL_TAIL", "1")
    # (with two levels of chunk elimination in front of it this 1024-block boundary system carries most of the
    # operator's conditioning, and set-up's accuracy check sends it to the register-blocked tail: the check is
    # switched off here, the point being the kernel's indexing of strided input)
    monkeypatch.setenv("AGGMG_CR_PCR_GUARD", "0")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 2**16, q, False, True)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(900)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for rank, err, scale, nex, chunked in sorted(q.get() for _ in range(2)):
        assert chunked
        # (measured 2.3e-8: what the unexamined parallel tail costs on this system -- the reason set-up examines it;
        # an indexing error would show at the scale of the iterate)
        assert err <= 1e-6 * scale, (rank, err, scale)


def test_four_ranks_overlapped_interface_exchange():
    """the next cycle's x0 interface exchange issued on a second stream under the split fine-level
    ascent (aggmg_vcycle_up_split_dev): same bits as the single-GPU run"""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, 2**16, q, True)) for r in range(4)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(900)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for rank, err, scale, nex, chunked in sorted(q.get() for _ in range(4)):
        # cycle 1: x0 + boundary system + coarse ghosts + prefetch; cycles 2, 3: three each
        assert chunked and nex == 10
        assert err == 0.0, (rank, err, scale)


@pytest.mark.parametrize("world,n,overlap,nex,p2p", [(2, 2048, False, 6, True), (2, 2**16, False, 9, True),
                                                     (4, 2**16, False, 9, True), (4, 2**16, True, 10, True),
                                                     (3, 3 * 2**14, True, None, True), (4, 2**20, True, 10, True),
                                                     (4, 2**16, True, 10, False), (2, 2048, False, 6, False)])
def test_native_schedule_matches_single_gpu(world, n, overlap, nex, p2p):
    """aggmg_dist_vcycle_dev (the schedule in C++, csrc/dist.hip) with its collectives routed through gloo:
    2 / 3 / 4 ranks on the one GPU, replicated and chunked coarsest solve, with and without the overlapped
    interface exchange; interface exchanges as neighbour messages straight between the vectors (grouped
    send / recv, the default) or as pack -> all-gather -> unpack (AGGMG_DIST_P2P=0); the chunk-boundary system
    gathered in place -- owned values bitwise those of the single-GPU cycle, same number of exchanges
    as the Python schedule"""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, overlap, True, "blockJac", p2p)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(900)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for rank, err, scale, got_nex, chunked in sorted(q.get() for _ in range(world)):
        assert err == 0.0, (rank, err, scale)
        # (three ranks: 2^14 coarsest blocks per rank are whole chunks, but the plan's chunk size decides)
        assert nex is None or (got_nex == nex and chunked == (n >= 2**16)), (chunked, got_nex)
    # (2^20 elements on four ranks: 2^18 per rank, 2^14 coarsest blocks each -- the partitioned path at a size where every
    # rank runs many tiles per level and several chunks of the coarsest solve; the 8-rank 2^24 job itself needs 8 GPUs)


def test_block_gauss_seidel_partitioned():
    """the labelled extension (red-black block Gauss-Seidel) under element partitioning: ghost
    layers twice as deep per sweep, colours by the parity of the element index (local ranges start on even
    elements); 2 and 4 ranks on the one GPU, library schedule, bitwise against the single-GPU Gauss-Seidel
    cycle of the same library"""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    for world, overlap in ((2, False), (4, True)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, 2**16, q, overlap, True, "blockGS")) for r in range(world)]
        for pr in procs:
            pr.start()
        for pr in procs:
            pr.join(900)
        assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
        for rank, err, scale, nex, chunked in sorted(q.get() for _ in range(world)):
            assert err == 0.0, (world, rank, err, scale)


def _cg_worker(rank, world, port, n, ps, sweeps, q, p2p=True, smoother="jac"):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import torch
    import torch.distributed as dist
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformCgDgHierarchy, build_device_cg_hierarchy
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ["AGGMG_DIST_P2P"] = "1" if p2p else "0"
    try:
        nPre, nPost = sweeps
        ctx = mg.Context(0)
        comm = D.Comm(world, rank, staged=True)
        alpha = 1.0 if smoother == "blockGS" else 2.0 / 3.0
        layout = D.CgRankLayout(n, ps, world, rank, nPre, nPost, smoother=smoother)
        engine, U = D.build_local_cg(n, ps, layout, ctx, comm)
        assert engine.H.level_kinds() == ['fused_chain'] * len(ps) + ['coarsest']
        dv = D.NativeDistributedVCycle(engine, layout, comm, collectives="torch")
        b = torch.from_numpy(U.rhs()).to(engine.dev)
        x = engine.new(layout.local_dofs(0))
        y = engine.new(layout.local_dofs(0))
        for _ in range(3):
            dv.vcycle(x, b, y, nPre, nPost, alpha, overlap_next=True)
            x, y = y, x
        torch.cuda.synchronize()
        got = x.cpu().numpy()[layout.owned_index(0)]
        Ug = UniformCgDgHierarchy(n, ps=ps)
        ctx2 = mg.Context(0)
        Hg = build_device_cg_hierarchy(Ug, ctx2, smoother=smoother)
        assert Hg.level_kinds() == ['fused_chain'] * len(ps) + ['coarsest']
        N = len(Ug.rhs())
        bg = ctx2.to_device(Ug.rhs())
        xa, xb = ctx2.to_device(np.zeros(N)), ctx2.alloc(N)
        for _ in range(3):
            Hg.vcycle_dev(xa, bg, xb, nPre, nPost, alpha)
            xa, xb = xb, xa
        ref = xa.download()[layout.global_index(0)]
        q.put((rank, float(np.max(np.abs(got - ref))), float(np.max(np.abs(ref))), dv.exchanges, dv.chunked))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,ps,sweeps,p2p", [(2, 2048, (4, 2, 1), (3, 3), True), (4, 2**13, (4, 2, 1), (3, 3), True),
                                                  (2, 1024, (2, 1), (1, 2), True), (3, 3 * 700, (3,), (2, 2), True),
                                                  (4, 2**13, (4, 2, 1), (3, 3), False)])
def test_cg_chain_hierarchy_partitioned(world, n, ps, sweeps, p2p):
    """BASELINE config 5's shape under element partitioning: CG p-chain (chain kernels on every rank's
    sub-mesh, vertices-first local numbering, one shared vertex per interface) + DG p=0, the library's
    schedule with its collectives over gloo -- owned vertices and interior nodes bitwise those of the
    single-GPU cycle"""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_cg_worker, args=(r, world, port, n, ps, sweeps, q, p2p)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(900)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for rank, err, scale, nex, chunked in sorted(q.get() for _ in range(world)):
        assert err == 0.0, (rank, err, scale)


@pytest.mark.parametrize("smoother", ["addSchwarz", "hybridSchwarz", "blockGS"])
@pytest.mark.parametrize("world,n,ps", [(2, 2048, (4, 2, 1)), (4, 2**13, (4, 2, 1))])
def test_cg_chain_schwarz_and_element_gs_partitioned(world, n, ps, smoother):
    """the smoothers BASELINE config 5 and cg_smoother name beyond point-Jacobi, under element partitioning: additive
    and hybrid element Schwarz (src/smoother.jl:1-46,104-134: two elements of reach per sweep) and the red-black
    element Gauss-Seidel extension (four; local ranges start on even elements) with ghost layers sized by
    CgRankLayout(smoother=) -- 2 and 4 ranks on the one GPU, owned values bitwise those of the single-GPU cycle"""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_cg_worker, args=(r, world, port, n, ps, (3, 3), q, True, smoother)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(900)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for rank, err, scale, nex, chunked in sorted(q.get() for _ in range(world)):
        assert err == 0.0, (smoother, rank, err, scale)


def _rccl_in_library_worker(port, q):
    sys.path[:0] = [ROOT]
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        import agglomerationmultigrid1d_amd as mg
        from agglomerationmultigrid1d_amd import distributed as D
        n, ratios, p = 4096, (4, 2, 2), 3
        ctx = mg.Context(0)
        comm = D.Comm(1, 0)
        layout = D.RankLayout(n, ratios, [p + 1, 2, 2, 2], 1, 0)
        engine, U = D.build_local_uniform(n, p, 1, ratios, layout, ctx, comm)
        dv = D.NativeDistributedVCycle(engine, layout, comm, collectives="rccl")
        inp = torch.arange(640, dtype=torch.float64, device="cuda") * 0.25
        out = torch.zeros(640, dtype=torch.float64, device="cuda")
        dv.allgather(out, inp)                       # ncclAllGather issued by the library on its stream
        b = torch.from_numpy(U.rhs()).to(engine.dev)
        x, y = engine.new(layout.local_dofs(0)), engine.new(layout.local_dofs(0))
        for _ in range(2):
            dv.vcycle(x, b, y, overlap_next=True)
            x, y = y, x
        torch.cuda.synchronize()
        # the same two cycles replayed as hipGraphs (ncclAllGather captured): 2 eager + 2 captures + 4 replays
        x2, y2 = engine.new(layout.local_dofs(0)), engine.new(layout.local_dofs(0))
        for rep in range(4):
            x2.zero_(), y2.zero_()
            torch.cuda.synchronize()     # torch's stream and the library's own stream are not ordered otherwise
            dv.vcycle(x2, b, y2, overlap_next=True, graph=True)
            dv.vcycle(y2, b, x2, overlap_next=True, graph=True)
            torch.cuda.synchronize()
        ginfo = dv.graph_info()
        gerr = float(torch.max(torch.abs(x2 - x)).item())
        from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
        Hg = build_device_hierarchy(UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios), ctx)
        xa, xb, bg = ctx.to_device(np.zeros(4 * n)), ctx.alloc(4 * n), ctx.to_device(U.rhs())
        for _ in range(2):
            Hg.vcycle_dev(xa, bg, xb)
            xa, xb = xb, xa
        q.put((bool(torch.equal(out, inp)), dv.rccl_ranks, dv.info()["backend"],
               float(np.max(np.abs(x.cpu().numpy() - xa.download()))), ginfo, gerr))
    finally:
        dist.destroy_process_group()


def test_rccl_inside_the_library_single_rank():
    """aggmg_rccl_unique_id / aggmg_dist_init_rccl / ncclAllGather from C++ (librccl loaded with dlopen, the copy
    torch already holds) on the one rank a single-GPU box allows, and the native cycle on top of it"""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_in_library_worker, args=(free_port(), q))
    pr.start()
    pr.join(600)
    assert pr.exitcode == 0
    same, ranks, backend, err, ginfo, gerr = q.get()
    assert same and ranks == 1 and backend == "rccl" and err == 0.0
    # graph replay gives the same bits; if RCCL could not be captured the library stays eager and says so
    assert gerr == 0.0
    assert (ginfo["captured"] == 2 and ginfo["replays"] >= 4) or ginfo["broken"], ginfo


def _nccl_worker(port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        sys.path[:0] = [ROOT]
        from agglomerationmultigrid1d_amd import distributed as D
        inp = torch.arange(640, dtype=torch.float64, device="cuda") * 0.5
        out = torch.empty(640, dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(out, inp)         # the call Comm.all_gather makes for world > 1
        dist.barrier()
        comm = D.Comm(1, 0)
        comm.world = 2                                 # take the collective branch of Comm.max
        m = comm.max(3.25)
        torch.cuda.synchronize()
        q.put((bool(torch.equal(out, inp)), m, dist.get_backend()))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_smoke_single_rank():
    """RCCL through torch.distributed (backend "nccl") initialises on the box and runs the fp64
    device-tensor collectives the partitioned driver issues -- with the one rank a single-GPU box
    allows; the multi-rank exchange pattern itself is covered by the gloo tests above."""
    import torch.multiprocessing as mp
    from test_distributed_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_nccl_worker, args=(free_port(), q))
    pr.start()
    pr.join(600)
    assert pr.exitcode == 0
    same, m, backend = q.get()
    assert same and m == 3.25 and backend == "nccl"


@pytest.mark.parametrize("dist_config,launcher,smoother", [(4, "torchrun", "jac"), (5, "torchrun", "jac"), (4, "bare", "jac"),
                                                           (5, "bare", "hybridSchwarz"), (5, "torchrun", "blockGS")])
def test_bench_two_ranks_under_torch_distributed_run(dist_config, launcher, smoother, tmp_path):
    """bench.py --gpus 2 as the driver launches it (python -m torch.distributed.run, one process per rank) and in the
    bare form `python bench.py --gpus 2` (bench.py starts the ranks itself as a child process), with both
    ranks sharing this box's GPU over gloo: the N > 1 path end to end -- rank-local generators, library set-up,
    partitioned cycles, max-over-ranks timing, ONE JSON line from rank 0"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AGGMG_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    port = 29600 + dist_config + 10 * ["jac", "hybridSchwarz", "blockGS"].index(smoother)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
            "--log2-elems", "16", "--cg-log2-elems", "16", "--dist-config", str(dist_config), "--dist-smoother", smoother]
    if launcher == "bare":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["metric"] == "fine_level_dof_updates_per_s_per_vcycle" and d["value"] > 0 and d["vs_baseline"] is None
    assert d["config"]["backend"] == "gloo" and "roofline" in d


def _thread_ranks(world, fn):
    """fn(rank, comm) on `world` ThreadComm ranks: threads of THIS process, each with its own library context on GPU 0
    (the box admits six GPU processes; the north-star job has eight ranks)"""
    import threading
    from agglomerationmultigrid1d_amd import distributed as D
    g = D.ThreadGroup(world)
    out, errs = [None] * world, []

    def one(r):
        try:
            out[r] = fn(r, D.ThreadComm(g, r))
        except BaseException as exc:
            import traceback
            errs.append((r, traceback.format_exc()))
            g.barrier.abort()

    ts = [threading.Thread(target=one, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(900)
    assert not errs, errs[0][1]
    return out


@pytest.mark.parametrize("n,chunk_log2,coarse_overlap", [(2**16, None, 0), (2**19, None, 0), (2**16, 1, 0), (2**19, None, 1), (2**16, None, 1)])
def test_eight_ranks_config4_match_single_gpu(n, chunk_log2, coarse_overlap, monkeypatch):
    """BASELINE config 4 at the north-star world size: the config-3 hierarchy element-partitioned over EIGHT ranks (threads
    of one process sharing the GPU, the library's C++ schedule with host-staged collectives): first / last rank one-sided,
    six interior ranks with two neighbours, chunk-interleaved boundary system gathered from eight ranks.  Owned values after
    three cycles are bitwise those of the single-GPU cycle.  2^16 and 2^19 fine elements: 2^12 / 2^15 coarsest blocks, chunked
    elimination with the default chunk size of a partitioned run; AGGMG_DIST_COARSE_CHUNK_LOG2 = 1: the smallest
    chunks there are (two blocks): a gathered boundary system of thousands of rows through the register-blocked
    tail."""
    import torch
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    if chunk_log2 is not None:
        monkeypatch.setenv("AGGMG_DIST_COARSE_CHUNK_LOG2", str(chunk_log2))
    # coarse_overlap: the exchange of the coarsest solution's ghost blocks on the side stream under the middle tiles of the
    # two-level ascent, the end tiles after it (aggmg_vcycle_up_coarse_dev) -- the same bits
    monkeypatch.setenv("AGGMG_DIST_COARSE_OVERLAP", str(coarse_overlap))
    world, ratios, p = 8, (4, 2, 2), 3
    Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
    ctx2 = mg.Context(0)
    Hg = build_device_hierarchy(Ug, ctx2)
    bg = ctx2.to_device(Ug.rhs())
    xa, xb = ctx2.to_device(np.zeros(len(Ug.rhs()))), ctx2.alloc(len(Ug.rhs()))
    for _ in range(3):
        Hg.vcycle_dev(xa, bg, xb)
        xa, xb = xb, xa
    ref = xa.download()
    Hg.free()

    def rank_fn(rank, comm):
        ctx = mg.Context(0)
        layout = D.RankLayout(n, ratios, [p + 1, 2, 2, 2], world, rank)
        engine, U = D.build_local_uniform(n, p, 1, ratios, layout, ctx, comm)
        assert all(engine.H.structured_levels()) and engine.Hc.coarse_info()['on_device']
        dv = D.NativeDistributedVCycle(engine, layout, comm, collectives="torch")
        b = torch.from_numpy(U.rhs()).to(engine.dev)
        x, y = engine.new(layout.local_dofs(0)), engine.new(layout.local_dofs(0))
        for _ in range(3):
            dv.vcycle(x, b, y, overlap_next=True)
            x, y = y, x
        torch.cuda.synchronize()
        got = x.cpu().numpy()[layout.owned_slice(0)]
        lo, hi = layout.own[0]
        ref_own = ref[lo * (p + 1):hi * (p + 1)]
        res = (float(np.max(np.abs(got - ref_own))), float(np.max(np.abs(ref_own))), dv.chunked, dv.exchanges)
        comm.barrier()
        dv.free()
        return res

    res = _thread_ranks(world, rank_fn)
    for rank, (err, scale, chunked, nex) in enumerate(res):
        assert chunked, rank                                    # 4096 coarsest blocks and more: every rank eliminates its own chunks
        if chunk_log2 is None:
            assert err == 0.0, (rank, err, scale)
        else:       # the reference run plans its own chunks and tail: equal to the accuracy of the coarsest solves (measured
            # 3.6e-8 after three cycles: cond(A_c) eps; an indexing error would show at the scale of the iterate)
            assert err <= 1e-6 * scale, (rank, err, scale)


@pytest.mark.parametrize("n,smoother", [(2**14, "jac"), (2**14, "blockGS")])
def test_eight_ranks_config5_match_single_gpu(n, smoother):
    """BASELINE config 5's shape (CG p = 4 -> 2 -> 1 -> DG p = 0) over EIGHT ranks (threads sharing the GPU): point-Jacobi
    and the block-GS extension config 5 names, owned vertices and interior nodes bitwise those of the single-GPU cycle"""
    import torch
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformCgDgHierarchy, build_device_cg_hierarchy
    world, ps = 8, (4, 2, 1)
    alpha = 1.0 if smoother == "blockGS" else 2.0 / 3.0
    Ug = UniformCgDgHierarchy(n, ps=ps)
    ctx2 = mg.Context(0)
    Hg = build_device_cg_hierarchy(Ug, ctx2, smoother=smoother)
    N = len(Ug.rhs())
    bg = ctx2.to_device(Ug.rhs())
    xa, xb = ctx2.to_device(np.zeros(N)), ctx2.alloc(N)
    for _ in range(3):
        Hg.vcycle_dev(xa, bg, xb, 3, 3, alpha)
        xa, xb = xb, xa
    ref = xa.download()
    Hg.free()

    def rank_fn(rank, comm):
        ctx = mg.Context(0)
        layout = D.CgRankLayout(n, ps, world, rank, 3, 3, smoother=smoother)
        engine, U = D.build_local_cg(n, ps, layout, ctx, comm)
        assert engine.H.level_kinds() == ['fused_chain'] * len(ps) + ['coarsest']
        dv = D.NativeDistributedVCycle(engine, layout, comm, collectives="torch")
        b = torch.from_numpy(U.rhs()).to(engine.dev)
        x, y = engine.new(layout.local_dofs(0)), engine.new(layout.local_dofs(0))
        for _ in range(3):
            dv.vcycle(x, b, y, 3, 3, alpha, overlap_next=True)
            x, y = y, x
        torch.cuda.synchronize()
        got = x.cpu().numpy()[layout.owned_index(0)]
        want = ref[layout.global_index(0)]
        comm.barrier()
        dv.free()
        return float(np.max(np.abs(got - want))), float(np.max(np.abs(want)))

    for rank, (err, scale) in enumerate(_thread_ranks(world, rank_fn)):
        assert err == 0.0, (rank, err, scale)


@pytest.mark.parametrize("dist_config", [4, 5])
def test_bench_eight_ranks_rehearsal(dist_config):
    """`python bench.py --gpus 8 --rehearse-threads`: bench_main's N-rank code path end to end with eight ranks (threads of
    one process on this box's GPU) -- rank-local generators, library set-up, partitioned cycles, max-over-ranks timing, ONE
    JSON line from rank 0 with n_gpus 8.  (The driver's `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`
    differs in the process group only: tested with two processes in test_bench_two_ranks_under_torch_distributed_run.)"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--rehearse-threads", "--steps", "3", "--warmup", "1",
           "--log2-elems", "18", "--cg-log2-elems", "16", "--dist-config", str(dist_config)]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["backend"] == "threads" and "rehearsal" in d and d["roofline"]["frac"] <= 1.0
    o_ = d["outer_solvers_to_1e-8"]["multigrid"]      # the partitioned multigrid() loop ran and met its tolerance
    assert 0 < o_["cycles"] < 200 and o_["final_residual"] > 0


@pytest.mark.parametrize("world,n,p,ratios,sweeps", [(8, 2**15, 2, (2, 2, 2), (2, 1)), (6, 6 * 2**12, 3, (4, 2), (3, 3)),
                                                    (5, 5 * 2**12, 1, (4, 4, 2), (1, 1)), (8, 2**17, 3, (4, 2, 2), (4, 4))])
def test_thread_ranks_other_shapes_match_single_gpu(world, n, p, ratios, sweeps):
    """the partitioned cycle away from the benchmark shape: other degrees, ratios, level counts, sweep counts (ghost layers
    sized from them) and world sizes that are not powers of two -- ranks as threads sharing the GPU, the library's schedule
    with the coarse ghost exchange in its default place, owned values bitwise those of the single-GPU cycle"""
    import torch
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    nPre, nPost = sweeps
    alpha = 0.8
    Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
    ctx2 = mg.Context(0)
    Hg = build_device_hierarchy(Ug, ctx2)
    bg = ctx2.to_device(Ug.rhs())
    xa, xb = ctx2.to_device(np.zeros(len(Ug.rhs()))), ctx2.alloc(len(Ug.rhs()))
    for _ in range(3):
        Hg.vcycle_dev(xa, bg, xb, nPre, nPost, alpha)
        xa, xb = xb, xa
    ref = xa.download()
    Hg.free()

    def rank_fn(rank, comm):
        ctx = mg.Context(0)
        layout = D.RankLayout(n, ratios, [p + 1] + [2] * len(ratios), world, rank, nPre, nPost)
        engine, U = D.build_local_uniform(n, p, 1, ratios, layout, ctx, comm)
        dv = D.NativeDistributedVCycle(engine, layout, comm, collectives="torch")
        b = torch.from_numpy(U.rhs()).to(engine.dev)
        x, y = engine.new(layout.local_dofs(0)), engine.new(layout.local_dofs(0))
        for _ in range(3):
            dv.vcycle(x, b, y, nPre, nPost, alpha, overlap_next=True)
            x, y = y, x
        torch.cuda.synchronize()
        got = x.cpu().numpy()[layout.owned_slice(0)]
        lo, hi = layout.own[0]
        ref_own = ref[lo * (p + 1):hi * (p + 1)]
        res = (float(np.max(np.abs(got - ref_own))), float(np.max(np.abs(ref_own))), dv.chunked)
        comm.barrier()
        dv.free()
        return res

    for rank, (err, scale, chunked) in enumerate(_thread_ranks(world, rank_fn)):
        # (a replicated coarsest solve and the single-GPU one take the same launches: bitwise; a chunked one plans its chunks
        # for the partition: equal to the accuracy of the coarsest solves)
        assert err == 0.0 or (chunked and err <= 1e-9 * scale), (rank, err, scale, chunked)


@pytest.mark.parametrize("config,check_every", [(4, 1), (4, 3), (5, 1)])
def test_eight_ranks_partitioned_multigrid_loop_matches_single_gpu(config, check_every):
    """distributed.multigrid -- the loop of multigrid (src/solvers.jl:116-139) over the partitioned cycle of the library's
    C++ schedule, ||A x - b|| summed over the ranks' owned rows in rank order -- on EIGHT ranks (threads sharing the GPU)
    against aggmg_multigrid_dev on one GPU: the same cycle count on every rank, the same residual history on every rank
    bit for bit and equal to the single-GPU one to the order of the sums, owned iterates bitwise the single-GPU iterate.
    Config 4 (DG / agglomerated, block-Jacobi) and config 5's shape (CG chain, point-Jacobi: the owned DoFs are two
    index ranges, vertices and element-interior nodes)."""
    import torch
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import (UniformCgDgHierarchy, UniformDgAggHierarchy, build_device_cg_hierarchy,
                                                      build_device_hierarchy)
    world, maxiter = 8, 30
    ctx2 = mg.Context(0)
    if config == 4:
        n, p, ratios = 2**16, 3, (4, 2, 2)
        Ug = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios)
        Hg = build_device_hierarchy(Ug, ctx2)
    else:
        n, ps = 2**14, (4, 2, 1)
        Ug = UniformCgDgHierarchy(n, ps=ps)
        Hg = build_device_cg_hierarchy(Ug, ctx2)
    N = len(Ug.rhs())
    bg = ctx2.to_device(Ug.rhs())
    # a tolerance met after a handful of cycles: taken from the single-GPU history itself
    _, _, hist = mg.multigrid_dev(Hg, ctx2.to_device(np.zeros(N)), bg, 12, 1e-30)
    tol = 0.5 * (hist[6] + hist[7]) / np.linalg.norm(Ug.rhs())
    xg, itg, resg = mg.multigrid_dev(Hg, ctx2.to_device(np.zeros(N)), bg, maxiter, tol, check_every=check_every)
    assert itg == (8 if check_every == 1 else 9)
    ref = xg.download()
    Hg.free()

    def rank_fn(rank, comm):
        ctx = mg.Context(0)
        if config == 4:
            layout = D.RankLayout(n, ratios, [p + 1, 2, 2, 2], world, rank)
            engine, U = D.build_local_uniform(n, p, 1, ratios, layout, ctx, comm)
        else:
            layout = D.CgRankLayout(n, ps, world, rank, 3, 3)
            engine, U = D.build_local_cg(n, ps, layout, ctx, comm)
        dv = D.NativeDistributedVCycle(engine, layout, comm, collectives="torch")
        b = torch.from_numpy(U.rhs()).to(engine.dev)
        x, it, res = D.multigrid(dv, engine.new(layout.local_dofs(0)), b, maxiter, tol, check_every=check_every)
        torch.cuda.synchronize()
        if config == 4:
            lo, hi = layout.own[0]
            got, want = x.cpu().numpy()[layout.owned_slice(0)], ref[lo * (p + 1):hi * (p + 1)]
        else:
            got, want = x.cpu().numpy()[layout.owned_index(0)], ref[layout.global_index(0)]
        comm.barrier()
        dv.free()
        return it, res, float(np.max(np.abs(got - want)))

    out = _thread_ranks(world, rank_fn)
    for rank, (it, res, err) in enumerate(out):
        assert it == itg and res == out[0][1], (rank, it, itg)
        assert np.allclose(res, resg, rtol=1e-10, atol=1e-13 * np.linalg.norm(Ug.rhs())), (rank, res, resg)
        assert err == 0.0, (rank, err)
