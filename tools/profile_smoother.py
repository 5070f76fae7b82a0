#!/usr/bin/env python3
"""The command rocprofv3 profiles for BASELINE config 2 (DG n = 2^20, p = 3, block-Jacobi + residual on one GPU):
phases of identical launches, separated by a marker dispatch (copy_segments_kernel), on K independent copies of
the operator, smoother and vectors taken in turn -- K = 3 copies hold ~0.75 GB, three times the 256 MB Infinity
Cache, so every launch streams from HBM; K = 1 is the loop as `smoother_only` used to time it (the same 250 MB
swept again and again: largely served by the Infinity Cache).

    python tools/profile_smoother.py [--copies K] [--reps R]
    phases: 1 sweep per launch | residual | 4 sweeps per launch | 8 sweeps per launch
Summary: tools/summarize_smoother_profile.py."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = (("sweeps_1_per_launch", 1), ("residual", 0), ("sweeps_4_per_launch", 4), ("sweeps_8_per_launch", 8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=3)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--log2-elems", type=int, default=20)
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.api import _ptr
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    ctx = mg.Context(0)
    lib = ctx.lib
    U = UniformDgAggHierarchy(2 ** args.log2_elems, p=3, pAgg=1, ratios=())
    A = U.stiffness_csc(0)
    N = A.shape[0]
    sets = []
    for _ in range(args.copies):
        op = mg.DeviceOperator(A, _lib.OP_STIFFNESS, ctx)
        S = mg.BlockJacobi(op, U.descriptor(0).mBlockInds, ctx)
        sets.append((op, S, ctx.to_device(np.zeros(N)), ctx.alloc(N), ctx.to_device(U.rhs()), ctx.alloc(N)))
    mk_src, mk_dst = ctx.alloc(64), ctx.alloc(64)
    P, I = ctypes.c_void_p * 1, ctypes.c_int64 * 1

    def marker():
        ctx.check(lib.aggmg_copy_segments_dev(ctx.handle, 1, P(mk_src.ptr.value), P(mk_dst.ptr.value), I(1), I(64), I(64), I(64)))

    for name, per_launch in PHASES:
        marker()
        for i in range(args.reps):
            op, S, u, v, b, r = sets[i % args.copies]
            if per_launch:
                ctx.check(lib.aggmg_smooth_dev(ctx.handle, op.handle, S.handle, _ptr(u), _ptr(b), 2.0 / 3.0, per_launch, _ptr(v)))
            else:
                ctx.check(lib.aggmg_residual_dev(ctx.handle, op.handle, _ptr(u), _ptr(b), _ptr(r)))
    marker()
    ctx.synchronize()
    print(json.dumps({"N": N, "nnz": int(A.nnz), "copies": args.copies, "reps": args.reps, "phases": [p[0] for p in PHASES]}))


if __name__ == "__main__":
    main()
