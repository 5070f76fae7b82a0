"""Diagnostic: shader-clock stamps inside the cyclic-reduction kernels (workgroup 0, thread 0).
Needs the stamp build:  hipcc ... -DAGGMG_CR_STAMPS -o build_variants/libaggmg_stamps.so
    AGGMG_HIP_LIB=build_variants/libaggmg_stamps.so python tools/exp_cr_stamps.py --log2-blocks 15
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-blocks", type=int, default=15)
    args = ap.parse_args()
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd import api as mg
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, block_tridiag_to_csc, _csc
    nb = 2 ** args.log2_blocks
    n = nb * 16
    ratios = (4, 2, 2)
    E = 16 * 8
    Ul = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=ratios, elem_range=(0, E))
    Ur = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=ratios, elem_range=(n - E, n))
    nc = 3
    mid = [x[4] for x in Ul.levels[nc]['A']]
    g = [np.broadcast_to(b, (nb,) + b.shape).copy() for b in mid]
    for i in range(3):
        g[i][:4] = Ul.levels[nc]['A'][i][:4]
        g[i][-4:] = Ur.levels[nc]['A'][i][-4:]
    colptr, rowval, nzval, N = block_tridiag_to_csc(*g)
    ctx = mg.Context(0)
    Ac = mg.DeviceOperator(_csc(colptr, rowval, nzval, (N, N)), _lib.OP_STIFFNESS, ctx)
    Hc = mg.MeshHierarchy(None, [Ac], [], [], ctx=ctx, keep_host=False, coarse_mode=_lib.COARSE_AUTO)
    print("coarse info", Hc.coarse_info())
    rng = np.random.default_rng(0)
    b = ctx.to_device(rng.standard_normal(N))
    z = ctx.to_device(np.zeros(N))
    x = ctx.alloc(N)
    for _ in range(20):
        Hc.vcycle_dev(z, b, x, 0, 0, 1.0)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        Hc.vcycle_dev(z, b, x, 0, 0, 1.0)
    ctx.synchronize()
    print("us per solve", (time.perf_counter() - t0) / 200 * 1e6)
    lib = ctx.lib
    if hasattr(lib, "aggmg_debug_cr_stamps"):
        out = (ctypes.c_ulonglong * 96)()
        lib.aggmg_debug_cr_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
        assert lib.aggmg_debug_cr_stamps(out) == 0
        s = np.array(out[:], dtype=np.int64).reshape(3, 32)
        for k, name in enumerate(["chunk_forward", "chunk_backward", "tail"]):
            v = s[k][s[k] > 0]
            print(name, "stamps (cycles since first):", (v - v[0]).tolist() if len(v) else [])


if __name__ == "__main__":
    main()
