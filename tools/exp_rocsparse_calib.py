"""Calibration of the generic CSR kernels against the vendor library: rocSPARSE's csrmv (y = A x; with and without its
analysis step) on the SAME operators the generic kernels of libaggmg_hip.so are timed on in bench.py (config 2's DG p = 3
operator at 2^20 elements; the CG p = 4 operator at 2^20 elements), next to the library's generic residual kernel
(r = b - A u: one vector more to read).  Says what the CSR format with int32 indices reaches on this machine, i.e. what
the "generic fallback" rows of DESIGN.md section 4 can be held against.  Measurement aid; rocSPARSE is NOT used by the
product.
    python tools/exp_rocsparse_calib.py
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import scipy.sparse as sp
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import UniformCgDgHierarchy, UniformDgAggHierarchy
    ctx = mg.Context(0)
    rs = ctypes.CDLL("/opt/rocm/lib/librocsparse.so")
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7")      # (the runtime the library already loaded: rocSPARSE runs on the null stream)
    handle, descr = ctypes.c_void_p(), ctypes.c_void_p()
    assert rs.rocsparse_create_handle(ctypes.byref(handle)) == 0
    assert rs.rocsparse_create_mat_descr(ctypes.byref(descr)) == 0
    out = {}
    for name in ("dg_p3_2p20", "cg_p4_2p20"):
        if name.startswith("dg"):
            U = UniformDgAggHierarchy(2 ** 20, p=3, pAgg=1, ratios=(4,))
            A = U.stiffness_csc(0)
        else:
            U = UniformCgDgHierarchy(2 ** 20, ps=(4,))
            A = U.A[0]
        A = sp.csr_matrix(A)
        A.sort_indices()
        N, nnz = A.shape[0], A.nnz
        val = ctx.to_device(A.data.astype(np.float64))
        # int32 index arrays ride in float64 device vectors (plumbing: the Context only hands out double vectors)
        def dev_i32(a):
            a = np.ascontiguousarray(a, dtype=np.int32)
            pad = np.zeros((a.size + 1) // 2 * 2, dtype=np.int32)
            pad[:a.size] = a
            return ctx.to_device(pad.view(np.float64))
        rp, ci = dev_i32(A.indptr), dev_i32(A.indices)
        x = ctx.to_device(np.random.default_rng(0).standard_normal(N))
        y = ctx.alloc(N)
        one, zero = ctypes.c_double(1.0), ctypes.c_double(0.0)
        alg_bytes = 12 * nnz + 4 * (N + 1) + 16 * N       # values + columns, row pointer, read x, write y
        res = {"N": N, "nnz": nnz, "algorithmic_bytes_spmv": alg_bytes}
        for analysed in (False, True):
            info = ctypes.c_void_p()
            assert rs.rocsparse_create_mat_info(ctypes.byref(info)) == 0
            if analysed:
                st = rs.rocsparse_dcsrmv_analysis(handle, 111, N, N, nnz, descr, val.ptr, rp.ptr, ci.ptr, info)
                assert st == 0, st

            def run():
                st = rs.rocsparse_dcsrmv(handle, 111, N, N, nnz, ctypes.byref(one), descr, val.ptr, rp.ptr, ci.ptr,
                                         info if analysed else None, x.ptr, ctypes.byref(zero), y.ptr)
                assert st == 0, st
            for _ in range(5):
                run()
            ctx.synchronize()
            hip.hipDeviceSynchronize()
            t0 = time.perf_counter()
            reps = 200
            for _ in range(reps):
                run()
            hip.hipDeviceSynchronize()
            us = 1e6 * (time.perf_counter() - t0) / reps
            key = "rocsparse_csrmv_adaptive" if analysed else "rocsparse_csrmv_no_analysis"
            res[key] = {"us": us, "algorithmic_GBs": alg_bytes / us / 1e3, "frac_of_8TBs": alg_bytes / us / 1e3 / 8000.0}
            # the result is what scipy computes
            if analysed:
                hip.hipDeviceSynchronize()
                yh = y.download()
                ref = A @ x.download()
                res["max_rel_err_vs_scipy"] = float(np.max(np.abs(yh - ref)) / np.max(np.abs(ref)))
            rs.rocsparse_destroy_mat_info(info)
        # the library's generic residual on the same operator (no structure hints: the smoother is never built)
        op = mg.DeviceOperator(sp.csc_matrix(A), _lib.OP_STIFFNESS, ctx)
        b = ctx.to_device(np.zeros(N))
        r = ctx.alloc(N)
        call = lambda: ctx.check(ctx.lib.aggmg_residual_dev(ctx.handle, op.handle, x.ptr, b.ptr, r.ptr))
        for _ in range(5):
            call()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            call()
        ctx.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / 200
        rb = alg_bytes + 8 * N
        res["libaggmg_generic_residual"] = {"us": us, "algorithmic_GBs": rb / us / 1e3, "frac_of_8TBs": rb / us / 1e3 / 8000.0,
                                            "algorithmic_bytes": rb}
        out[name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
