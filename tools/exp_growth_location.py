"""Where does the growing component of the 2^24 V-cycle iteration live, and is it seeded by
stale memory (non-zero output from x0 = 0, b = 0) or by arithmetic round-off?"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, default=24)
    ap.add_argument("--cycles", type=int, default=56)
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    n = 2 ** args.log2_elems
    U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=(4, 2, 2))
    ctx = mg.Context(0)
    H = build_device_hierarchy(U, ctx)
    N = 4 * n
    zero = ctx.to_device(np.zeros(N))
    # (1) after real work has left data in every internal buffer, iterate from exact zeros
    x, y = ctx.to_device(np.random.default_rng(1).standard_normal(N)), ctx.alloc(N)
    for _ in range(3):
        H.vcycle_dev(x, zero, y)
        x, y = y, x
    x, y = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    for _ in range(4):
        H.vcycle_dev(x, zero, y)
        x, y = y, x
    out = x.download()
    print("x0 = 0, b = 0 after 4 cycles (internal buffers dirty): max |x| =", float(np.max(np.abs(out))),
          " non-zeros:", int(np.count_nonzero(out)), flush=True)
    # (2) location of the growing component
    x, y = ctx.to_device(np.random.default_rng(1).standard_normal(N)), ctx.alloc(N)
    for _ in range(args.cycles):
        H.vcycle_dev(x, zero, y)
        x, y = y, x
    v = x.download()
    a = np.abs(v)
    top = np.argsort(a)[-12:][::-1]
    print("after", args.cycles, "cycles: max |x| =", float(a.max()), " median |x| =", float(np.median(a)))
    print("largest entries: row, element, element from the right end, row % 4, element % 16, value")
    for r in top:
        e = int(r) // 4
        print(f"   {int(r):>10d} {e:>9d} {n - 1 - e:>9d} {int(r) % 4} {e % 16:>2d} {v[r]: .3e}")
    # profile of |x| per element over the domain (64 bins) and near the two ends
    ea = a.reshape(n, 4).max(axis=1)
    bins = ea.reshape(64, -1).max(axis=1)
    print("max |x| per 1/64 of the domain:", " ".join(f"{b:.1e}" for b in bins))
    print("first 8 elements:", " ".join(f"{b:.1e}" for b in ea[:8]), " last 8:", " ".join(f"{b:.1e}" for b in ea[-8:]))


if __name__ == "__main__":
    main()
