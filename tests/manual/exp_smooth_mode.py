"""Amplification of the smoothest mode (element-wise constant quarter cosine, the lowest
eigenfunction of the Neumann-Dirichlet problem) by one V(3,3) cycle with b = 0:
  * the HIP path (fused block-tridiagonal kernels), and
  * the plain-C restatement in the reference's own operation order (oracle/aggmg_oracle_c.c,
    explicit r = rhs - A u, y = S \\ r, explicit L' r, banded LU), OpenMP over the host cores.
In exact arithmetic a two-grid-exact coarse correction removes this mode; what is left is the
floating-point error of computing its residual (lambda_min ~ 2.5 against entries ~ CDir n)."""
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
    ap.add_argument("--log2-elems", type=int, nargs="+", default=[20, 22, 23, 24])
    ap.add_argument("--cycles", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cdir-times-n", type=float, default=0.0,
                    help="choose CDir so that CDir * n has this value (default: the model problem's CDir = 1000 n)")
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    import c_oracle
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(min(16, os.cpu_count() or 1))
    except OSError:
        pass
    for E in args.log2_elems:
        n = 2 ** E
        cdir = args.cdir_times_n / n if args.cdir_times_n else 1000.0 * n
        U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=(4, 2, 2), CDir=cdir)
        N = 4 * n
        xc = (np.arange(n) + 0.5) / n
        mode_vec = np.repeat(np.cos(0.5 * np.pi * xc), 4)
        zero_h = np.zeros(N)
        line = f"2^{E}: CDir*n = {cdir * n:.2e}"
        # HIP path, both forms of the restricted residual
        from agglomerationmultigrid1d_amd import _lib
        ctx = mg.Context(0)
        H = build_device_hierarchy(U, ctx, keep_host=False)
        for name, mode in (("explicit", _lib.RESTRICT_EXPLICIT), ("preconditioned", _lib.RESTRICT_PRECONDITIONED)):
            try:
                H.set_restriction(mode)
            except mg.UnsupportedError:        # refused above _lib.RESTRICT_PRECONDITIONED_MAX_ELEMS fine elements
                line += f"  | HIP {name}: refused at this size"
                continue
            x, y, z = ctx.to_device(mode_vec), ctx.alloc(N), ctx.to_device(zero_h)
            f = []
            prev = np.linalg.norm(mode_vec)
            for _ in range(args.cycles):
                H.vcycle_dev(x, z, y)
                x, y = y, x
                cur = mg.norm2(x)
                f.append(cur / prev)
                prev = cur
            line += f"  | HIP {name}: " + " ".join(f"{v:.3f}" for v in f)
        H.free()
        if not args.no_cpu:
            As = [U.stiffness_csc(k) for k in range(U.nlevels)]
            Ls = [U.interpolation_csc(k) for k in range(U.nlevels - 1)]
            C = c_oracle.COracleHierarchy(As, Ls, [U.levels[k]['m'] for k in range(U.nlevels - 1)])
            C.enable_omp(As, Ls)
            xo = mode_vec.copy()
            f = []
            prev = np.linalg.norm(xo)
            for _ in range(args.cycles):
                xo, _, _ = C.vcycle_omp(xo, zero_h)
                cur = np.linalg.norm(xo)
                f.append(cur / prev)
                prev = cur
            line += "  | C restatement (reference order): " + " ".join(f"{v:.3f}" for v in f)
            del C, As, Ls
        print(line, flush=True)
        del U, H, ctx


if __name__ == "__main__":
    main()
