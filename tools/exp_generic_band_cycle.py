"""V(3,3) on a hierarchy whose levels run the GENERIC kernels: config 3's operators (DG p = 3 -> agglomerated levels) smoothed
by point-Jacobi (dg_smoother(mesh, A, :jac), src/smoother.jl:146-151) -- no block smoother, so no block-tridiagonal form; the
operators are banded, so the sweeps (and, on the descent, the residual) of a level are one csr_band_kernel launch.  Per-launch
times by kind.  Measurement aid.
    python tools/exp_generic_band_cycle.py --log2-elems 20
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, default=20)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    ctx = mg.Context(0)
    U = UniformDgAggHierarchy(2 ** args.log2_elems, p=3, pAgg=1, ratios=(4, 2, 2))
    nl = U.nlevels
    ops = [mg.DeviceOperator(U.stiffness_csc(k), _lib.OP_STIFFNESS, ctx) for k in range(nl)]
    sms = [mg.JacobiSmoother(ops[k], ctx) for k in range(nl - 1)]
    Ls = [mg.DeviceOperator(U.interpolation_csc(k), _lib.OP_TRANSFER, ctx) for k in range(nl - 1)]
    H = mg.MeshHierarchy(None, ops, sms, Ls, ctx=ctx, keep_host=False)
    b = ctx.to_device(U.rhs())
    N = len(U.rhs())
    xa, xb = ctx.alloc(N), ctx.alloc(N)
    for _ in range(3):
        H.vcycle_dev(xa, b, xb, 3, 3, 0.5)
        xa, xb = xb, xa
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        H.vcycle_dev(xa, b, xb, 3, 3, 0.5)
        xa, xb = xb, xa
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile_enable(1)
    for _ in range(5):
        H.vcycle_dev(xa, b, xb, 3, 3, 0.5)
        xa, xb = xb, xa
    ctx.synchronize()
    ctx.profile_enable(0)
    prof = ctx.profile_collect()
    print(json.dumps({"log2_elems": args.log2_elems, "level_kinds": H.level_kinds(), "ms_per_cycle": 1e3 * dt / args.steps,
                      "kernels_ms": {f"{k}_L{l}": [round(v[0] / v[1], 4), v[1]] for (k, l), v in sorted(prof.items())}}))


if __name__ == "__main__":
    main()
