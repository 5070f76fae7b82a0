"""The host-pointer entry point (aggmg_vcycle: what `multigrid_v_cycle(H, x0, b)` on host arrays calls) against the
device-resident one: PCIe-inclusive time per cycle, with the caller's arrays page-locked for the call (default) and
staged as pageable memory (AGGMG_HOST_REGISTER_MIN_BYTES=0).  Measurement aid.

    python tools/exp_host_entry.py --log2-elems 22
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
    ap.add_argument("--log2-elems", type=int, default=22)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    ctx = mg.Context(0)
    U = UniformDgAggHierarchy(2 ** args.log2_elems, p=3, pAgg=1, ratios=(4, 2, 2))
    H = build_device_hierarchy(U, ctx)
    b = U.rhs()
    N = len(b)
    x0 = np.zeros(N)
    x = mg.multigrid_v_cycle(H, x0, b)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = mg.multigrid_v_cycle(H, x, b)
    dt_host = (time.perf_counter() - t0) / args.steps
    # the same calls on arrays page-locked once (Context.pin / pinned_empty: aggmg_host_register / aggmg_host_alloc) and a
    # result array that is reused: three DMA transfers per call, nothing staged
    xp, bp, yp = ctx.pinned_empty(N), ctx.pin(b.copy()), ctx.pinned_empty(N)
    xp[:] = 0.0
    mg.multigrid_v_cycle(H, xp, bp, out=yp)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mg.multigrid_v_cycle(H, xp, bp, out=yp)
        xp, yp = yp, xp
    dt_pin = (time.perf_counter() - t0) / args.steps
    bd, xa, xb = ctx.to_device(b), ctx.to_device(x0), ctx.alloc(N)
    H.vcycle_dev(xa, bd, xb)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        H.vcycle_dev(xa, bd, xb)
        xa, xb = xb, xa
    ctx.synchronize()
    dt_dev = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"log2_elems": args.log2_elems, "N": N, "register_min_bytes": os.environ.get("AGGMG_HOST_REGISTER_MIN_BYTES", "default"),
                      "ms_per_cycle_host_pointers": 1e3 * dt_host, "ms_per_cycle_host_pointers_pinned": 1e3 * dt_pin,
                      "pcie_GBs_effective_pinned": 3 * 8 * N / dt_pin / 1e9,
                      "ms_per_cycle_device_resident": 1e3 * dt_dev,
                      "pcie_GBs_effective": 3 * 8 * N / dt_host / 1e9,
                      "dof_updates_per_s_host_pointers": 6 * N / dt_host}))


if __name__ == "__main__":
    main()
