"""Rehearse ONE rank's share of a W-rank run of bench.py on a single GPU.

The collectives are replaced by device-local loop-back copies of the same sizes (so the iterate is
meaningless), everything else -- the local hierarchy with its halos, the chunked coarsest solve,
the packing / unpacking around the exchanges, all launches -- is what rank R of W executes.  The
per-cycle time is therefore the rank-local cost (kernels + launch overhead) that RCCL latency adds
to on the multi-GPU node.  Measurement aid only.

    python tools/exp_dist_rank.py --log2-elems 24 --world 8 --rank 3
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class LoopbackComm:
    staged = False

    def __init__(self, world, rank):
        import torch
        self.torch = torch
        self.world, self.rank = world, rank
        self.calls = 0

    def all_gather(self, out, inp):
        out.view(self.world, -1).copy_(inp)
        self.calls += 1

    def barrier(self):
        pass

    def max(self, v):
        return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, default=24)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=4, choices=(4, 5),
                    help="4: the config-3 DG / agglomerated hierarchy partitioned; 5: CG p = 4 -> 2 -> 1 -> DG p = 0 (point-Jacobi)")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--graph", action="store_true", help="AGGMG_DIST_GRAPH: replay the cycle as a hipGraph")
    ap.add_argument("--cr-trace", action="store_true",
                    help="the coarsest solve's in-kernel timeline of the last cycle (needs AGGMG_HIP_LIB=build_trace/libaggmg_hip_trace.so)")
    ap.add_argument("--python-schedule", action="store_true",
                    help="the Python schedule (one ctypes / torch call per launch) instead of aggmg_dist_vcycle_dev")
    args = ap.parse_args()
    import torch
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd import api as mg
    from agglomerationmultigrid1d_amd import distributed as D
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, block_tridiag_to_csc, _csc

    n, p, ratios = 2 ** args.log2_elems, 3, (4, 2, 2)
    nPre = nPost = 3
    alpha = 2.0 / 3.0
    ctx = mg.Context(0)
    comm = LoopbackComm(args.world, args.rank)
    if args.config == 5:
        from agglomerationmultigrid1d_amd.uniform import UniformCgDgHierarchy
        ps = (4, 2, 1)
        layout = D.CgRankLayout(n, ps, args.world, args.rank, nPre, nPost)
        U = UniformCgDgHierarchy(n, ps=ps, elem_range=layout.loc[0])
        nl = U.nlevels
        ops = [mg.DeviceOperator(A, _lib.OP_STIFFNESS, ctx) for A in U.A]
        sms = [mg.JacobiSmoother(ops[k], ctx, U.element_nodes(k)) for k in range(nl - 1)]
        Ls = [mg.DeviceOperator(L, _lib.OP_TRANSFER, ctx) for L in U.L]
        H = mg.MeshHierarchy(None, ops, sms, Ls, ctx=ctx, keep_host=False, coarse_mode=_lib.COARSE_EXTERNAL)
        # global coarsest (DG p = 0) operator: interior rows alike on the uniform mesh, the ends from edge hierarchies
        E = 64
        Ul = UniformCgDgHierarchy(n, ps=ps, elem_range=(0, E))
        Ur = UniformCgDgHierarchy(n, ps=ps, elem_range=(n - E, n))
        nb = layout.ne[nl - 1]
        mid = [x[len(x) // 2] for x in U.dg0.levels[0]['A']]
        g = [np.broadcast_to(b_, (nb,) + b_.shape).copy() for b_ in mid]
        for i in range(3):
            g[i][:4] = Ul.dg0.levels[0]['A'][i][:4]
            g[i][-4:] = Ur.dg0.levels[0]['A'][i][-4:]
        p = 4      # (for the DoF count of the projected rate below: 4 n + 1 fine DoFs)
    else:
        layout = D.RankLayout(n, ratios, [p + 1, 2, 2, 2], args.world, args.rank, nPre, nPost)
        lo, hi = layout.loc[0]
        U = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios, elem_range=(lo, hi))
        nl = U.nlevels
        ops, sms = [], []
        for k in range(nl):
            op = mg.DeviceOperator(U.stiffness_csc(k), _lib.OP_STIFFNESS, ctx)
            ops.append(op)
            if k < nl - 1:
                sms.append(mg.BlockJacobi(op, U.descriptor(k).mBlockInds, ctx))
        Ls = [mg.DeviceOperator(U.interpolation_csc(k), _lib.OP_TRANSFER, ctx) for k in range(nl - 1)]
        H = mg.MeshHierarchy([U.descriptor(k) for k in range(nl)], ops, sms, Ls, ctx=ctx, keep_host=False,
                             coarse_mode=_lib.COARSE_EXTERNAL)
        # global coarsest operator: interior block rows are all alike on the uniform mesh; the first and
        # last few come from edge hierarchies
        nc = nl - 1
        E = 16 * 8
        Ul = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios, elem_range=(0, E))
        Ur = UniformDgAggHierarchy(n, p=p, pAgg=1, ratios=ratios, elem_range=(n - E, n))
        nb = layout.ne[nc]
        mid = [x[len(x) // 2] for x in U.levels[nc]['A']]
        g = [np.broadcast_to(b, (nb,) + b.shape).copy() for b in mid]
        for i in range(3):
            g[i][:4] = Ul.levels[nc]['A'][i][:4]
            g[i][-4:] = Ur.levels[nc]['A'][i][-4:]
    colptr, rowval, nzval, N = block_tridiag_to_csc(*g)
    Ac = mg.DeviceOperator(_csc(colptr, rowval, nzval, (N, N)), _lib.OP_STIFFNESS, ctx)
    Hc = D._replicated_coarse_hierarchy(Ac, ctx, args.world)     # as build_local_uniform does
    engine = D.HipEngine(H, Hc, ctx)
    if args.python_schedule:
        dv = D.DistributedVCycle(engine, layout, comm)
    else:   # the schedule inside the library with its device-local loop-back "all-gather"
        dv = D.NativeDistributedVCycle(engine, layout, comm, collectives="loopback")
    b = torch.from_numpy(U.rhs()).to(engine.dev)
    xa = engine.new(layout.local_dofs(0))
    xb = engine.new(layout.local_dofs(0))
    src, dst = xa, xb
    kw = dict(overlap_next=True)
    if args.graph:
        kw["graph"] = True
        args.warmup = max(args.warmup, 8)      # eager, capture, then replays
    for _ in range(args.warmup):
        dv.vcycle(src, b, dst, nPre, nPost, alpha, **kw)
        src, dst = dst, src
    torch.cuda.synchronize()
    if args.profile:
        ctx.profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dv.vcycle(src, b, dst, nPre, nPost, alpha, **kw)
        src, dst = dst, src
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"config": args.config, "world": args.world, "rank": args.rank, "log2_elems": args.log2_elems, "chunked": dv.chunked,
           "W": layout.W, "ms_per_cycle": 1e3 * dt / args.steps, "ms_issue_per_cycle": 1e3 * t_issue / args.steps,
           "schedule": "python" if args.python_schedule else "library (aggmg_dist_vcycle_dev)",
           "collectives_per_cycle": (comm.calls if args.python_schedule else dv.exchanges) // (args.steps + args.warmup),
           "projected_value_if_all_ranks_alike": n * (p + 1) * (nPre + nPost) * args.steps / dt}
    if args.graph:
        out["graph"] = dv.graph_info()
    if args.profile:
        ctx.profile_enable(0)
        prof = ctx.profile_collect()
        out["kernels_ms"] = {f"{k}_L{l}": v[0] / v[1] for (k, l), v in sorted(prof.items())}
        out["kernels_sum_ms"] = sum(v[0] / v[1] for v in prof.values())
    print(json.dumps(out), flush=True)
    if args.cr_trace:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import cr_trace
        fn = cr_trace.trace_fn(ctx)
        fn(ctx.handle, None, 1)
        dv.vcycle(src, b, dst, nPre, nPost, alpha, **kw)
        torch.cuda.synchronize()
        cr_trace.report(cr_trace.fetch(ctx, fn))


if __name__ == "__main__":
    main()
