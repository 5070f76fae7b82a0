"""multigrid()'s loop at the reference's semantics -- a residual check after EVERY cycle (src/solvers.jl:124-131) -- on the
device: ms per iteration against the bare cycle.  Measurement aid.
    python tools/exp_outer_loop.py --log2-elems 24"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, default=24)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--cg", action="store_true", help="config 5's shape (CG p = 4 -> 2 -> 1 -> DG p = 0, point-Jacobi) instead of config 3's")
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd.uniform import (UniformCgDgHierarchy, UniformDgAggHierarchy, build_device_cg_hierarchy,
                                                      build_device_hierarchy)
    ctx = mg.Context(0)
    if args.cg:
        U = UniformCgDgHierarchy(2 ** args.log2_elems, ps=(4, 2, 1))
        H = build_device_cg_hierarchy(U, ctx)
    else:
        U = UniformDgAggHierarchy(2 ** args.log2_elems, p=3, pAgg=1, ratios=(4, 2, 2))
        H = build_device_hierarchy(U, ctx)
    b = ctx.to_device(U.rhs())
    N = len(U.rhs())
    x0 = ctx.to_device(np.zeros(N))
    out = {"log2_elems": args.log2_elems, "hierarchy": "cg p=4,2,1 -> dg p=0" if args.cg else "dg p=3 -> agg 4:1, 2:1, 2:1"}
    for ce in (1, 2, 8):
        mg.multigrid_dev(H, x0, b, 2 * ce, 1e-30, check_every=ce)
        ctx.synchronize()
        t0 = time.perf_counter()
        _, ncyc, res = mg.multigrid_dev(H, x0, b, args.iters * ce, 1e-30, check_every=ce)
        ctx.synchronize()
        out[f"ms_per_cycle_check_every_{ce}"] = 1e3 * (time.perf_counter() - t0) / ncyc
        out[f"res_{ce}"] = res[-1]
    if args.profile:   # per-launch times (HIP events) of a run with a check after every cycle
        ctx.profile_enable(1)
        mg.multigrid_dev(H, x0, b, args.iters, 1e-30, check_every=1)
        ctx.profile_enable(0)
        out["kernels_ms_check_every_1"] = {f"{k}_L{l}": [round(v[0] / v[1], 4), v[1]] for (k, l), v in sorted(ctx.profile_collect().items())}
        ctx.profile_enable(1)
        mg.multigrid_dev(H, x0, b, args.iters, 1e-30, check_every=args.iters)
        ctx.profile_enable(0)
        out["kernels_ms_no_check"] = {f"{k}_L{l}": [round(v[0] / v[1], 4), v[1]] for (k, l), v in sorted(ctx.profile_collect().items())}
    xa, xb = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    H.vcycle_dev(xa, b, xb)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        H.vcycle_dev(xa, b, xb)
        xa, xb = xb, xa
    ctx.synchronize()
    out["ms_per_bare_cycle"] = 1e3 * (time.perf_counter() - t0) / args.iters
    # iterative_smoother_solve (src/solvers.jl:189-213), a test after every sweep
    A, S = H._ops[0], H.mSmoothers[0]
    for ce in (1, 7):
        mg.smoother_solve_dev(A, S, x0, b, 14, 1e-30, 2.0 / 3.0, check_every=ce)
        ctx.synchronize()
        t0 = time.perf_counter()
        _, nit, res, _ = mg.smoother_solve_dev(A, S, x0, b, 7 * args.iters, 1e-30, 2.0 / 3.0, check_every=ce)
        ctx.synchronize()
        out[f"ms_per_sweep_check_every_{ce}"] = 1e3 * (time.perf_counter() - t0) / nit
        out[f"sweep_res_{ce}"] = res[-1]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
