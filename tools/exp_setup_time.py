"""Where the hierarchy set-up time goes (host generation vs. library set-up vs. upload)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, default=22)
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    ctx = mg.Context(0)
    T = {}

    def tic(name, t0):
        ctx.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0

    t0 = time.perf_counter()
    U = UniformDgAggHierarchy(2 ** args.log2_elems, p=3, pAgg=1, ratios=(4, 2, 2))
    tic("generate blocks (numpy)", t0)
    n = U.nlevels
    ops, sms, Ls = [], [], []
    for k in range(n):
        t0 = time.perf_counter()
        csc = U.stiffness_csc(k)
        tic("blocks -> CSC (numpy)", t0)
        t0 = time.perf_counter()
        op = mg.DeviceOperator(csc, _lib.OP_STIFFNESS, ctx)
        tic("aggmg_csc_upload stiffness", t0)
        ops.append(op)
        if k < n - 1:
            t0 = time.perf_counter()
            d = U.descriptor(k)
            tic("descriptor / mBlockInds (numpy)", t0)
            t0 = time.perf_counter()
            sms.append(mg.BlockJacobi(op, d.mBlockInds, ctx))
            tic("aggmg_blockjacobi_setup", t0)
    for k in range(n - 1):
        t0 = time.perf_counter()
        csc = U.interpolation_csc(k)
        tic("interpolation -> CSC (numpy)", t0)
        t0 = time.perf_counter()
        Ls.append(mg.DeviceOperator(csc, _lib.OP_TRANSFER, ctx))
        tic("aggmg_csc_upload transfer", t0)
    t0 = time.perf_counter()
    H = mg.MeshHierarchy([U.descriptor(k) for k in range(n)], ops, sms, Ls, ctx=ctx, keep_host=False)
    tic("aggmg_hier_create (transfers, cyclic-reduction factors)", t0)
    for k, v in T.items():
        print(f"{v:8.2f} s  {k}")
    print(f"{sum(T.values()):8.2f} s  total")


if __name__ == "__main__":
    main()
