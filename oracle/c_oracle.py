"""ctypes front-end of oracle/liboracle_c.so (the plain-C restatement).  TEST INFRASTRUCTURE:
used by tests/, smoke() and bench.py's cpu_baseline leg only."""
import ctypes
import os
import time
from ctypes import POINTER, Structure, c_double, c_int, c_int32, c_int64, c_void_p

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))


class csc_t(Structure):
    _fields_ = [("n_rows", c_int64), ("n_cols", c_int64), ("colptr", c_void_p), ("rowval", c_void_p),
                ("nzval", c_void_p)]


class smoother_t(Structure):
    _fields_ = [("kind", c_int), ("m", c_int64), ("nb", c_int64), ("diag", c_void_p), ("lu", c_void_p),
                ("piv", c_void_p)]


class csr_t(Structure):
    _fields_ = [("n_rows", c_int64), ("rowptr", c_void_p), ("colind", c_void_p), ("val", c_void_p)]


class banded_t(Structure):
    _fields_ = [("n", c_int64), ("kl", c_int), ("ku", c_int), ("ldab", c_int), ("ab", c_void_p),
                ("ipiv", c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle_c.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE])
        _lib = ctypes.CDLL(path)
        _lib.oc_extract_factor_blocks.restype = c_int64
    return _lib


class COracleHierarchy:
    """Holds the arrays of a hierarchy in the layout the C restatement reads.  Build from lists of
    SciPy CSC stiffness / interpolation matrices and per-level block sizes (0 -> point Jacobi)."""

    def __init__(self, stiffness, interpolation, block_sizes):
        L = lib()
        self.n = len(stiffness)
        self._keep = []
        self.A = (csc_t * self.n)()
        self.Lm = (csc_t * max(self.n - 1, 1))()
        self.S = (smoother_t * max(self.n - 1, 1))()
        for k, M in enumerate(stiffness):
            self.A[k] = self._csc(M)
        for k, M in enumerate(interpolation):
            self.Lm[k] = self._csc(M)
        for k in range(self.n - 1):
            N = stiffness[k].shape[0]
            m = block_sizes[k]
            s = smoother_t()
            if m == 0:
                d = np.ascontiguousarray(sp.csc_matrix(stiffness[k]).diagonal(), dtype=np.float64)
                self._keep.append(d)
                s.kind, s.m, s.nb, s.diag = 0, 1, N, d.ctypes.data
            else:
                nb = N // m
                lu = np.zeros(nb * m * m)
                piv = np.zeros(nb * m, dtype=np.int32)
                st = L.oc_extract_factor_blocks(ctypes.byref(self.A[k]), c_int64(m), c_int64(nb),
                                                lu.ctypes.data_as(c_void_p), piv.ctypes.data_as(c_void_p))
                if st != 0:
                    raise np.linalg.LinAlgError(f"singular block {st}")
                self._keep += [lu, piv]
                s.kind, s.m, s.nb, s.lu, s.piv = 1, m, nb, lu.ctypes.data, piv.ctypes.data
            self.S[k] = s
        self.coarse = banded_t()
        st = L.oc_banded_factor(ctypes.byref(self.A[self.n - 1]), ctypes.byref(self.coarse))
        if st != 0:
            raise np.linalg.LinAlgError("coarsest operator singular / out of memory")
        self.N = [M.shape[0] for M in stiffness]
        self.block_sizes = [int(m) for m in block_sizes]
        self.work = np.zeros(5 * sum(self.N))

    def _csc(self, M):
        M = sp.csc_matrix(M)
        M.sort_indices()
        cp = np.ascontiguousarray(M.indptr, dtype=np.int64)
        rv = np.ascontiguousarray(M.indices, dtype=np.int64)
        nz = np.ascontiguousarray(M.data, dtype=np.float64)
        self._keep += [cp, rv, nz]
        c = csc_t()
        c.n_rows, c.n_cols = M.shape
        c.colptr, c.rowval, c.nzval = cp.ctypes.data, rv.ctypes.data, nz.ctypes.data
        return c

    def vcycle(self, x0, b, nPre=3, nPost=3, alpha=2.0 / 3.0):
        """-> (x, seconds_total, seconds_in_coarsest_solve)"""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.empty(self.N[0])
        cs = c_double(0.0)
        t0 = time.perf_counter()
        st = lib().oc_vcycle(c_int(self.n), self.A, self.S, self.Lm, ctypes.byref(self.coarse),
                             x0.ctypes.data_as(c_void_p), b.ctypes.data_as(c_void_p), c_int(nPre), c_int(nPost),
                             c_double(alpha), out.ctypes.data_as(c_void_p), self.work.ctypes.data_as(c_void_p),
                             ctypes.byref(cs))
        dt = time.perf_counter() - t0
        assert st == 0
        return out, dt, cs.value

    def _csr(self, M):
        M = sp.csr_matrix(M)
        M.sort_indices()
        rp = np.ascontiguousarray(M.indptr, dtype=np.int64)
        ci = np.ascontiguousarray(M.indices, dtype=np.int64)
        v = np.ascontiguousarray(M.data, dtype=np.float64)
        self._keep += [rp, ci, v]
        c = csr_t()
        c.n_rows = M.shape[0]
        c.rowptr, c.colind, c.val = rp.ctypes.data, ci.ctypes.data, v.ctypes.data
        return c

    def enable_omp(self, stiffness, interpolation):
        """row-gather (CSR) copies for the OpenMP variant"""
        self.Ar = (csr_t * self.n)()
        self.Lr = (csr_t * max(self.n - 1, 1))()
        for k, M in enumerate(stiffness):
            self.Ar[k] = self._csr(M)
        for k, M in enumerate(interpolation):
            self.Lr[k] = self._csr(M)

    def vcycle_omp(self, x0, b, nPre=3, nPost=3, alpha=2.0 / 3.0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.empty(self.N[0])
        cs = c_double(0.0)
        t0 = time.perf_counter()
        st = lib().oc_vcycle_omp(c_int(self.n), self.Ar, self.S, self.Lm, self.Lr, ctypes.byref(self.coarse),
                                 x0.ctypes.data_as(c_void_p), b.ctypes.data_as(c_void_p), c_int(nPre), c_int(nPost),
                                 c_double(alpha), out.ctypes.data_as(c_void_p), self.work.ctypes.data_as(c_void_p),
                                 ctypes.byref(cs))
        dt = time.perf_counter() - t0
        assert st == 0
        return out, dt, cs.value

    def smooth(self, k, u, b, alpha, nsweeps):
        u = np.array(u, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        work = np.zeros(3 * self.N[k])
        t0 = time.perf_counter()
        lib().oc_smooth(ctypes.byref(self.A[k]), ctypes.byref(self.S[k]), b.ctypes.data_as(c_void_p),
                        c_double(alpha), c_int(nsweeps), u.ctypes.data_as(c_void_p), work.ctypes.data_as(c_void_p))
        return u, time.perf_counter() - t0

    def __del__(self):
        try:
            lib().oc_banded_free(ctypes.byref(self.coarse))
        except Exception:
            pass


def from_oracle_hierarchy(H):
    """COracleHierarchy of an aggmg_oracle.MeshHierarchy (block-Jacobi levels must have contiguous
    blocks, which DG / agglomerated levels do)."""
    ms = []
    for S in H.mSmoothers[:-1] if len(H.mSmoothers) == len(H.mStiffness) else H.mSmoothers:
        ms.append(S.mBlockInds.shape[0] if hasattr(S, "mBlockInds") else 0)
    return COracleHierarchy(H.mStiffness, H.mInterpolation, ms)
