"""CG p-chain hierarchy (BASELINE config 5 shape) on one GPU: set-up breakdown, V-cycle time, per-kernel
HIP-event table, algorithmic bytes -- chain path vs the generic CSR path."""
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
    ap.add_argument("--ps", type=str, default="4,2,1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--generic", action="store_true", help="operators only (no element lists), chain detection off: generic CSR kernels")
    ap.add_argument("--detect", action="store_true", help="operators only; the library recognises the chain itself (AGGMG_OPT_DETECT_CHAIN)")
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import UniformCgDgHierarchy
    ctx = mg.Context(0)
    T = {}

    def tic(name, t0):
        ctx.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0

    ps = tuple(int(x) for x in args.ps.split(","))
    n = 2 ** args.log2_elems
    t0 = time.perf_counter()
    U = UniformCgDgHierarchy(n, ps=ps)
    tic("generate CSC (numpy)", t0)
    ops, sms, Ls = [], [], []
    for k in range(U.nlevels):
        t0 = time.perf_counter()
        ops.append(mg.DeviceOperator(U.A[k], _lib.OP_STIFFNESS, ctx))
        tic("aggmg_csc_upload stiffness", t0)
        if k < U.nlevels - 1:
            t0 = time.perf_counter()
            el = None if (args.generic or args.detect) else U.element_nodes(k)
            tic("element lists (numpy)", t0)
            t0 = time.perf_counter()
            sms.append(mg.JacobiSmoother(ops[k], ctx, el, detect=not args.generic))
            tic("aggmg_jacobi_setup_elements", t0)
    for k in range(U.nlevels - 1):
        t0 = time.perf_counter()
        Ls.append(mg.DeviceOperator(U.L[k], _lib.OP_TRANSFER, ctx))
        tic("aggmg_csc_upload transfer", t0)
    t0 = time.perf_counter()
    H = mg.MeshHierarchy(None, ops, sms, Ls, ctx=ctx, keep_host=False)
    tic("aggmg_hier_create", t0)
    for k, v in T.items():
        print(f"{v:8.2f} s  {k}", flush=True)
    print(f"{sum(T.values()):8.2f} s  total set-up", flush=True)
    print("level kinds:", H.level_kinds(), "coarse:", H.coarse_info(), flush=True)

    N = U.A[0].shape[0]
    bm = U.algorithmic_bytes(3, 3)
    b = ctx.to_device(U.rhs())
    xa, xb = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    for _ in range(2):
        H.vcycle_dev(xa, b, xb)
        xa, xb = xb, xa
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        H.vcycle_dev(xa, b, xb)
        xa, xb = xb, xa
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out_loop = ctx.alloc(N)
    H.vcycles_dev(xa, b, out_loop, args.steps)
    ctx.synchronize()
    t0 = time.perf_counter()
    H.vcycles_dev(xa, b, out_loop, args.steps)
    ctx.synchronize()
    dt_loop = (time.perf_counter() - t0) / args.steps
    ctx.profile_enable(1)
    for _ in range(args.steps):
        H.vcycle_dev(xa, b, xb)
        xa, xb = xb, xa
    ctx.synchronize()
    ctx.profile_enable(False)
    prof = ctx.profile_collect()
    vb = sum(l["vcycle"] for l in bm)
    out = {"n": n, "ps": ps, "N_fine": N, "ms_per_vcycle": 1e3 * dt, "ms_per_cycle_in_vcycles_loop": 1e3 * dt_loop, "dof_updates_per_s": N * 6 / dt,
           "algorithmic_GBs": vb / dt / 1e9, "frac_of_8TBs": vb / dt / 8e12,
           "kernels": {f"{k}_L{l}": round(v[0] / v[1], 4) for (k, l), v in sorted(prof.items())}}
    for k, lm in enumerate(bm):
        for kind, byts in (("fused_down", 3 * lm["sweep"] + lm["residual"] + lm["restrict"]),
                           ("fused_up", 3 * lm["sweep"] + lm["prolong"])):
            if (kind, k) in prof:
                ms = prof[(kind, k)][0] / prof[(kind, k)][1]
                out[f"{kind}_L{k}_algorithmic_TBs"] = round(byts / ms / 1e9, 2)
    # residual history of the multigrid loop (sanity: the cycle converges at this size)
    x, ncyc, res = mg.multigrid_dev(H, ctx.to_device(np.zeros(N)), b, 40, 1e-9, check_every=4)
    out["multigrid_res"] = [float(f"{r:.3e}") for r in res]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
