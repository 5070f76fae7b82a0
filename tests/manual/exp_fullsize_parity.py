"""One-off evidence run: the HIP V-cycle against the plain-C restatement in the reference's own
operation order (oracle/aggmg_oracle_c.c, OpenMP row-gather variant: same per-row summation
order) at sizes the test suite cannot afford, on a random iterate.  Reports the differences in
the norms the tolerances are stated in; asserts nothing."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, nargs="+", default=[20, 22])
    ap.add_argument("--cycles", type=int, default=3)
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    import aggmg_oracle as o
    import c_oracle
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(min(16, os.cpu_count() or 1))
    except OSError:
        pass
    for E in args.log2_elems:
        n = 2 ** E
        U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=(4, 2, 2))
        N = 4 * n
        b = U.rhs()
        x0 = o.splitmix_normal(N, 0)
        As = [U.stiffness_csc(k) for k in range(U.nlevels)]
        Ls = [U.interpolation_csc(k) for k in range(U.nlevels - 1)]
        A = As[0].tocsr()
        C = c_oracle.COracleHierarchy(As, Ls, [U.levels[k]['m'] for k in range(U.nlevels - 1)])
        C.enable_omp(As, Ls)
        ctx = mg.Context(0)
        H = build_device_hierarchy(U, ctx, keep_host=False)
        xc, xh = x0.copy(), x0.copy()
        db = ctx.to_device(b)
        xc_mode = np.repeat(np.cos(0.5 * np.pi * (np.arange(n) + 0.5) / n), 4)
        xc_mode /= np.linalg.norm(xc_mode)
        scale = max(np.linalg.norm(b), np.linalg.norm(b - A @ x0))
        print(f"2^{E}: ||b|| = {np.linalg.norm(b):.3e}, ||b - A x0|| = {np.linalg.norm(b - A @ x0):.3e}")
        for k in range(args.cycles):
            xc, _, _ = C.vcycle_omp(xc, b)
            dx, dy = ctx.to_device(xh), ctx.alloc(N)
            H.vcycle_dev(dx, db, dy)
            xh = dy.download()
            d = xh - xc
            smooth = float(d @ xc_mode)
            rest = d - smooth * xc_mode
            print(f"   cycle {k + 1}: ||x_hip - x_c|| / ||x_c|| = {np.linalg.norm(d) / np.linalg.norm(xc):.2e}"
                  f"  (smoothest-mode part {abs(smooth) / np.linalg.norm(xc):.2e}, rest {np.linalg.norm(rest) / np.linalg.norm(xc):.2e})"
                  f"   ||A (x_hip - x_c)|| / scale = {np.linalg.norm(A @ d) / scale:.2e}"
                  f"   residuals: hip {np.linalg.norm(b - A @ xh) / np.linalg.norm(b):.6e}  c {np.linalg.norm(b - A @ xc) / np.linalg.norm(b):.6e}",
                  flush=True)
        H.free()
        del C, As, Ls, A, U, H, ctx


if __name__ == "__main__":
    main()
