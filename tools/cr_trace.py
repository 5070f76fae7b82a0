"""Timeline of the coarsest solve's three launches from inside the kernels: constant-clock stamps (100 MHz) written by
thread 0 of every workgroup at the step boundaries of csrc/cr_kernels.hpp -- a library compiled with -DAGGMG_CR_TRACE
(tools/build_cr_trace.sh -> build_trace/libaggmg_hip_trace.so; the product library carries none of this).

Says where the 50 / 26 / 60 us of forward / tail / backward go: dispatch ramp (entry stamps of the workgroups against
the first one), the streaming step 0 (its loads, its arithmetic), the small dependent steps, the drain."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("AGGMG_HIP_LIB", os.path.join(ROOT, "build_trace", "libaggmg_hip_trace.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

KINDS = ["forward", "tail", "backward"]
WGS, SLOTS = 4096, 16
SLOT_NAMES = {
    "forward": {0: "entry", 1: "lds zeroed", 2: "step0 d loaded (thread 0)", 3: "step0 arithmetic done (thread 0)",
                4: "step0 barrier", 5: "step1 barrier", 6: "step2 barrier", 7: "step3 barrier", 15: "end (stores issued+landed)"},
    "tail": {0: "entry", 1: "lds zeroed", 2: "step0 d loaded", 3: "step0 arithmetic", 4: "fwd step0", 5: "fwd step1", 6: "fwd step2",
             7: "fwd step3", 12: "last block solved", 11: "bwd step3", 10: "bwd step2", 9: "bwd step1", 8: "bwd step0", 15: "end"},
    "backward": {0: "entry", 1: "lds zeroed", 2: "stack loaded", 11: "bwd step3", 10: "bwd step2", 9: "bwd step1", 8: "bwd step0 (stores landed)",
                 15: "end"},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", type=str, default="20:2")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--warm", action="store_true", help="no cache flush between the solves")
    args = ap.parse_args()
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import _lib
    from exp_coarse import block_tridiag
    ctx = mg.Context(0)
    fn = ctx.lib.aggmg_debug_cr_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    assert fn(ctx.handle, None, 1) == 0  # allocates the buffer
    lg, m = (int(v) for v in args.case.split(":"))
    nb = 1 << lg
    A = block_tridiag(nb, m)
    N = A.shape[0]
    op = mg.DeviceOperator(A, _lib.OP_STIFFNESS, ctx)
    H = mg.MeshHierarchy(None, [op], [], [], ctx=ctx, keep_host=False, coarse_mode=_lib.COARSE_DEVICE_CR)
    b = ctx.to_device(np.random.default_rng(1).standard_normal(N))
    x, z = ctx.alloc(N), ctx.to_device(np.zeros(N))
    nflush = 1 << 27
    big = [ctx.alloc(nflush), ctx.alloc(nflush)]
    P, I = ctypes.c_void_p * 1, ctypes.c_int64 * 1
    srcs, dsts = P(big[0].ptr.value), P(big[1].ptr.value)
    rows, cols, ld = I(1), I(nflush), I(nflush)
    for _ in range(3):
        H.vcycle_dev(z, b, x, 0, 0, 1.0)
    out = np.zeros((3, WGS, SLOTS), dtype=np.uint64)
    runs = []
    for _ in range(args.reps):
        fn(ctx.handle, None, 1)
        if not args.warm:
            ctx.check(ctx.lib.aggmg_copy_segments_dev(ctx.handle, 1, srcs, dsts, rows, cols, ld, ld))
        H.vcycle_dev(z, b, x, 0, 0, 1.0)
        fn(ctx.handle, out.ctypes.data_as(ctypes.c_void_p), 0)
        runs.append(out.astype(np.int64).copy())
    print(f"case {args.case}: N = {N}; stamps in us relative to the first forward workgroup's entry (last of {args.reps} solves)")
    report(runs[-1])
    spans = []
    for R in runs:
        t0 = R[0, :, 0][R[0, :, 0] > 0].min()
        spans.append({kind: [round(float((R[k, :, 0][R[k, :, 0] > 0].min() - t0) / 100.0), 2),
                             round(float((R[k].max() - t0) / 100.0), 2)] for k, kind in enumerate(KINDS) if (R[k, :, 0] > 0).any()})
    print("\nfirst entry / last stamp per launch, every solve:", json.dumps(spans))
    r = A @ x.download() - b.download()
    print("rel residual", float(np.linalg.norm(r) / np.linalg.norm(b.download())))


def trace_fn(ctx):
    fn = ctx.lib.aggmg_debug_cr_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    return fn


def fetch(ctx, fn, clear=0):
    out = np.zeros((3, WGS, SLOTS), dtype=np.uint64)
    fn(ctx.handle, out.ctypes.data_as(ctypes.c_void_p), clear)
    return out.astype(np.int64)


def report(T):
    t_first = T[0, :, 0][T[0, :, 0] > 0].min()
    for k, kind in enumerate(KINDS):
        S = T[k]
        live = S[:, 0] > 0
        nw = int(live.sum())
        if not nw:
            continue
        e = (S[live, 0] - t_first) / 100.0
        print(f"\n{kind}: {nw} workgroups traced; entry min {e.min():.2f} median {np.median(e):.2f} max {e.max():.2f} us")
        for slot in sorted(SLOT_NAMES[kind], key=lambda s: np.median(S[live, s]) if (S[live, s] > 0).any() else 0):
            v = S[live, slot]
            ok = v > 0
            if not ok.any():
                continue
            rel_entry = (v[ok] - S[live, 0][ok]) / 100.0
            absu = (v[ok] - t_first) / 100.0
            print(f"  {SLOT_NAMES[kind][slot]:38s} since own entry: min {rel_entry.min():6.2f} med {np.median(rel_entry):6.2f} "
                  f"max {rel_entry.max():6.2f} | absolute: min {absu.min():6.2f} med {np.median(absu):6.2f} max {absu.max():6.2f}")


if __name__ == "__main__":
    main()
