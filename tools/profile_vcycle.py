#!/usr/bin/env python3
"""The command rocprofv3 profiles for profiles/: W warm-up + K default V(3,3) cycles of ONE hierarchy and
nothing else on the GPU, so that every kernel dispatch of the trace belongs to a cycle and the roles
(fused descent / ascent per level, coarsest-solve kernels) can be read off the dispatch order.

    python tools/profile_vcycle.py --kind dg|cg|cggeneric --log2-elems E [--steps K] [--warmup W]

dg: BASELINE config 3/4 hierarchy (DG p=3 -> AggDG 4:1 -> 2:1 -> 2:1); cg: config 5 shape (CG p=4 -> 2 -> 1
-> DG p=0); cggeneric: the same operators handed over WITHOUT element lists, i.e. through the generic CSR kernels
(csr_stream_kernel / csr_row_kernel), the fallback of operators without structure.  Summaries: tools/summarize_profiles.py."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", choices=("dg", "cg", "cggeneric"), default="dg",
                    help="cggeneric: the config-5 operators without their element lists -- generic CSR kernels")
    ap.add_argument("--log2-elems", type=int, default=24)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cr-trace", action="store_true",
                    help="in-kernel timeline of the coarsest solve INSIDE the last cycle "
                         "(AGGMG_HIP_LIB=build_trace/libaggmg_hip_trace.so, tools/cr_trace.py)")
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import uniform
    ctx = mg.Context(0)
    n = 2 ** args.log2_elems
    if args.kind == "dg":
        U = uniform.UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=(4, 2, 2))
        H = uniform.build_device_hierarchy(U, ctx)
    else:
        U = uniform.UniformCgDgHierarchy(n, ps=(4, 2, 1))
        H = uniform.build_device_cg_hierarchy(U, ctx, chain=(args.kind == "cg"))
    b = ctx.to_device(U.rhs())
    N = len(U.rhs())
    xa, xb = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    for _ in range(args.warmup + args.steps):
        H.vcycle_dev(xa, b, xb)
        xa, xb = xb, xa
    ctx.synchronize()
    if args.cr_trace:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import cr_trace
        fn = cr_trace.trace_fn(ctx)
        fn(ctx.handle, None, 1)
        H.vcycle_dev(xa, b, xb)
        ctx.synchronize()
        cr_trace.report(cr_trace.fetch(ctx, fn))
    print(json.dumps({"kind": args.kind, "log2_elems": args.log2_elems, "steps": args.steps, "levels": H.level_kinds()}))


if __name__ == "__main__":
    main()
