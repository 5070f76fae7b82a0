"""Coarsest direct solve on its own (block cyclic reduction, csrc/cr_kernels.hpp): a one-level hierarchy's
V-cycle is the solve.  Times it for the systems the benchmarked hierarchies end in -- 2^20 blocks of 2
(config 3/4 at 2^24 elements), 2^24 scalar rows (config 5 at 2^24), and one rank's share of each."""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def block_tridiag(nb, m, seed=0):
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    rows, cols, vals = [], [], []
    for dr, dc, cnt, shift in ((0, 0, nb, 4.0 * m), (1, 0, nb - 1, 0.0), (0, 1, nb - 1, 0.0)):
        blk = rng.standard_normal((cnt, m, m)) + shift * np.eye(m)
        e = np.arange(cnt)
        rows.append(((e + dr)[:, None, None] * m + ii).ravel())
        cols.append(((e + dc)[:, None, None] * m + jj).ravel())
        vals.append(blk.ravel())
    return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nb * m, nb * m))


def calibrate(ctx, args):
    """cold streaming ceiling: for every (bytes, workgroups, mode) the median of `steps` launches, each after 2 GB
    have been streamed through the caches (as --cold does before every solve)"""
    import ctypes
    nflush = 1 << 27
    big = [ctx.alloc(nflush), ctx.alloc(nflush)]
    P, I = ctypes.c_void_p * 1, ctypes.c_int64 * 1
    srcs, dsts = P(big[0].ptr.value), P(big[1].ptr.value)
    rows, cols, ld = I(1), I(nflush), I(nflush)
    mbs = [int(v) for v in args.calib_mb.split(",")]
    nmax = max(mbs) * 1000 * 1000 // 8 + 2
    a, b = ctx.to_device(np.ones(nmax)), ctx.alloc(nmax)
    ms = ctypes.c_double(0.0)
    for mb in mbs:
        for wgs in (int(v) for v in args.calib_wgs.split(",")):
            for mode, name in ((1, "read"), (0, "copy")):
                # mode 0 moves nbytes in and nbytes out: give it half, so that `mb` is the traffic of either mode
                nbytes = (mb * 1000 * 1000 // (1 if mode else 2)) // 16 * 16
                t = []
                for _ in range(max(5, args.steps // 5)):
                    ctx.check(ctx.lib.aggmg_copy_segments_dev(ctx.handle, 1, srcs, dsts, rows, cols, ld, ld))
                    ctx.check(ctx.lib.aggmg_debug_stream_copy(ctx.handle, b.ptr, a.ptr, nbytes, wgs, mode, ctypes.byref(ms)))
                    t.append(ms.value)
                med = float(np.median(t))
                print(json.dumps({"calibrate": name, "traffic_MB": mb, "workgroups": wgs, "us": round(1e3 * med, 2),
                                  "TBps": round(mb * 1e6 / (med * 1e-3) / 1e12, 3), "min_us": round(1e3 * min(t), 2)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=str, default="24:1,20:2,21:1,17:2,18:2,14:1")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--cold", action="store_true",
                    help="stream 2 GB through the caches before every solve and time the solve with HIP events: "
                         "inside a V-cycle the factors come from HBM, not from the Infinity Cache")
    ap.add_argument("--calibrate", action="store_true",
                    help="instead of solves: a plain cold 16-byte streaming kernel (aggmg_debug_stream_copy) of the byte counts "
                         "the solve's streaming steps move, on the same grid (256 workgroups x 256 threads) and on larger "
                         "ones -- the ceiling those steps are held against")
    ap.add_argument("--calib-mb", type=str, default="82,135,256,1024")
    ap.add_argument("--calib-wgs", type=str, default="256,512,1024,4096")
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import _lib
    ctx = mg.Context(0)
    if args.calibrate:
        calibrate(ctx, args)
        return
    for case in args.cases.split(","):
        lg, m = (int(v) for v in case.split(":"))
        nb = 1 << lg
        A = block_tridiag(nb, m)
        N = A.shape[0]
        op = mg.DeviceOperator(A, _lib.OP_STIFFNESS, ctx)
        H = mg.MeshHierarchy(None, [op], [], [], ctx=ctx, keep_host=False, coarse_mode=_lib.COARSE_DEVICE_CR)
        b = ctx.to_device(np.random.default_rng(1).standard_normal(N))
        x, z = ctx.alloc(N), ctx.to_device(np.zeros(N))
        for _ in range(3):
            H.vcycle_dev(z, b, x, 0, 0, 1.0)
        ctx.synchronize()
        if args.cold:
            import ctypes
            nflush = 1 << 27
            big = [ctx.alloc(nflush), ctx.alloc(nflush)]
            P, I = ctypes.c_void_p * 1, ctypes.c_int64 * 1
            srcs, dsts = P(big[0].ptr.value), P(big[1].ptr.value)
            rows, cols, ld = I(1), I(nflush), I(nflush)
            ctx.profile_enable(1)
            for _ in range(args.steps):
                ctx.check(ctx.lib.aggmg_copy_segments_dev(ctx.handle, 1, srcs, dsts, rows, cols, ld, ld))
                H.vcycle_dev(z, b, x, 0, 0, 1.0)
            ctx.synchronize()
            ctx.profile_enable(False)
            prof = ctx.profile_collect()
            tot, cnt = prof[("coarse", 0)]
            ms = tot / cnt
            del big
        else:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                H.vcycle_dev(z, b, x, 0, 0, 1.0)
            ctx.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / args.steps
        r = A @ x.download() - b.download()
        print(json.dumps({"log2_blocks": lg, "m": m, "ms_per_solve": round(ms, 4), "rows_per_us": round(N / ms / 1e3, 1),
                          "rel_residual": float(np.linalg.norm(r) / np.linalg.norm(b.download()))}), flush=True)
        H.free()


if __name__ == "__main__":
    main()
