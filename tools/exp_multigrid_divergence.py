"""Does the stationary multigrid loop (src/solvers.jl:116-139) converge at 2^E fine elements, and
does the answer depend on the coarsest solver (device cyclic reduction vs host banded LU)?"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, default=24)
    ap.add_argument("--cycles", type=int, default=96)
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    n = 2 ** args.log2_elems
    U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=(4, 2, 2))
    ctx = mg.Context(0)
    nl = U.nlevels
    ops = [mg.DeviceOperator(U.stiffness_csc(k), _lib.OP_STIFFNESS, ctx) for k in range(nl)]
    Ls = [mg.DeviceOperator(U.interpolation_csc(k), _lib.OP_TRANSFER, ctx) for k in range(nl - 1)]
    desc = [U.descriptor(k) for k in range(nl)]
    sms = [mg.BlockJacobi(ops[k], desc[k].mBlockInds, ctx) for k in range(nl - 1)]
    b = U.rhs()
    nb = np.linalg.norm(b)
    N = len(b)
    db = ctx.to_device(b)
    for name, mode, keep in (("device cyclic reduction", _lib.COARSE_DEVICE_CR, True), ("host banded LU", _lib.COARSE_HOST_BANDED, False)):
        H = mg.MeshHierarchy(desc, ops, sms, Ls, ctx=ctx, keep_host=keep, coarse_mode=mode)
        _, ncyc, res = mg.multigrid_dev(H, ctx.to_device(np.zeros(N)), db, args.cycles, 0.0, check_every=8)
        print(f"2^{args.log2_elems} {name}: ||b||={nb:.3e} coarse={H.coarse_info()}")
        print("   ||A x - b|| / ||b|| every 8 cycles:", " ".join(f"{r / nb:.2e}" for r in res), flush=True)
        H.free()


if __name__ == "__main__":
    main()
