"""Power iteration on the V-cycle's error propagation E = I - V A: with b = 0 the loop of
multigrid() is x <- E x, so ||A x_k|| grows or decays like rho(E)^k once the dominant mode leads.
Random x0 (all modes present from the start), sizes 2^E fine elements."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, nargs="+", default=[20, 22, 23, 24])
    ap.add_argument("--cycles", type=int, default=64)
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    for E in args.log2_elems:
        n = 2 ** E
        U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=(4, 2, 2))
        ctx = mg.Context(0)
        H = build_device_hierarchy(U, ctx)
        N = 4 * n
        rng = np.random.default_rng(1)
        x0 = ctx.to_device(rng.standard_normal(N))
        zero = ctx.to_device(np.zeros(N))
        _, _, res = mg.multigrid_dev(H, x0, zero, args.cycles, 0.0, check_every=8)
        r = np.array(res)
        ratios = (r[1:] / r[:-1]) ** (1.0 / 8.0)
        print(f"2^{E}: ||A x_k|| every 8 cycles: " + " ".join(f"{v:.2e}" for v in r))
        print(f"      per-cycle factor between checks: " + " ".join(f"{v:.3f}" for v in ratios), flush=True)
        H.free()
        del H, U, x0, zero, ctx


if __name__ == "__main__":
    main()
