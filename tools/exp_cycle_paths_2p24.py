"""At 2^E fine elements, b = 0, random x0: ||A x_k|| for (a) separate aggmg_vcycle_dev calls,
(b) aggmg_vcycles_dev in groups of 8 (fused across cycles); and whether 8 fused cycles equal 8
separate ones bitwise (the property tests/test_gpu_fullsize.py asserts at 2^22)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, default=24)
    ap.add_argument("--cycles", type=int, default=48)
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    n = 2 ** args.log2_elems
    U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=(4, 2, 2))
    ctx = mg.Context(0)
    H = build_device_hierarchy(U, ctx)
    N = 4 * n
    x0h = np.random.default_rng(1).standard_normal(N)
    zero = ctx.to_device(np.zeros(N))
    A = H._ops[0]

    def anorm(dx):
        out = mg.api.ctypes.c_double(0.0)
        ctx.check(ctx.lib.aggmg_residual_norm_dev(ctx.handle, A.handle, dx.ptr, zero.ptr, mg.api.ctypes.byref(out)))
        return out.value

    # (a) separate cycles
    x, y = ctx.to_device(x0h), ctx.alloc(N)
    ra = []
    for k in range(args.cycles):
        H.vcycle_dev(x, zero, y)
        x, y = y, x
        if (k + 1) % 8 == 0:
            ra.append(anorm(x))
    xa = x.download()
    print("separate cycles  :", " ".join(f"{v:.2e}" for v in ra), flush=True)
    # (b) fused groups of 8
    x, y = ctx.to_device(x0h), ctx.alloc(N)
    rb = []
    for k in range(args.cycles // 8):
        H.vcycles_dev(x, zero, y, 8)
        x, y = y, x
        rb.append(anorm(x))
    xb = x.download()
    print("fused groups of 8:", " ".join(f"{v:.2e}" for v in rb), flush=True)
    print("bitwise equal after", args.cycles, "cycles:", bool(np.array_equal(xa, xb)),
          " max |diff| / max |x|:", float(np.max(np.abs(xa - xb)) / max(np.max(np.abs(xa)), 1e-300)))
    # first group only
    x, y, z = ctx.to_device(x0h), ctx.alloc(N), ctx.alloc(N)
    H.vcycles_dev(x, zero, y, 8)
    s = ctx.to_device(x0h)
    t = ctx.alloc(N)
    for _ in range(8):
        H.vcycle_dev(s, zero, t)
        s, t = t, s
    d = np.abs(y.download() - s.download())
    print("first 8 cycles: fused == separate bitwise:", bool(d.max() == 0.0), " max |diff|:", float(d.max()),
          " at row", int(d.argmax()), "of", N)


if __name__ == "__main__":
    main()
