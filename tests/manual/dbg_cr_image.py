"""debugging aid: the coarsest solve with the LDS-image path against SciPy, where it goes wrong"""
import os, sys
import numpy as np
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AGGMG_CR_PROBE"] = "0"
from test_gpu_coarse_cr import block_tridiag
import agglomerationmultigrid1d_amd as mg
from agglomerationmultigrid1d_amd import _lib

for nb, m in [(1 << 18, 1), (1 << 19, 2)]:
    A = block_tridiag(nb, m, seed=3)
    N = A.shape[0]
    ctx = mg.default_context()
    op = mg.DeviceOperator(A, _lib.OP_STIFFNESS, ctx)
    H = mg.MeshHierarchy(None, [op], [], [], ctx=ctx, keep_host=False, coarse_mode=_lib.COARSE_DEVICE_CR)
    q, nq, mm, nblk = [v for v in H.coarse_plan()] if hasattr(H, "coarse_plan") else (None,) * 4
    b = np.random.default_rng(5).standard_normal(N)
    bd, xd, z = ctx.to_device(b), ctx.alloc(N), ctx.to_device(np.zeros(N))
    H.vcycle_dev(z, bd, xd, 0, 0, 1.0)
    x = xd.download()
    ref = spla.splu(A).solve(b)
    err = np.abs(x - ref)
    bad = np.nonzero(err > 1e-9 * np.abs(ref).max())[0]
    print(f"nb={nb} m={m} q={q} max err {err.max():.3e}  bad rows {len(bad)}", end="")
    if len(bad):
        blk = bad // m
        e8 = [float(err[np.arange(k, min(N, 4096), 8)].max()) for k in range(8)]
        print("  err by block mod 8 (first 4096):", ["%.1e" % t for t in e8])
        print(f"  first bad blocks {blk[:8]} last {blk[-4:]}  mod 8: {sorted(set(blk % 8))}  mod 512 (first 10) {sorted(set(blk % 512))[:10]}"
              f" chunk ids {sorted(set(blk >> (q or 12)))[:6]}...")
    else:
        print()
    H.free()
