"""Host-side mirror of the reference's operator interface for the V-cycle hot path.

The reference is Julia (no FFI of its own; Julia is not available in the build image), so the
host side above the C ABI is Python and mirrors the reference names, argument meaning, return
shapes and error behaviour:

    reference (Julia)                                   here
    --------------------------------------------------  -------------------------------------
    apply_smoother(S, B; alpha)    src/smoother.jl:6-81   apply_smoother(S, B, alpha=1.0)
    dg_smoother(mesh, A, :blockJac|:jac)  :142-168        dg_smoother(mesh, A, 'blockJac'|'jac')
    cg_smoother(mesh, A, :jac|:addSchwarz|:hybridSchwarz) cg_smoother(mesh, A, ...)
                                   :88-139
    struct MeshHierarchy           src/mesh_heirarchy.jl:17-28   class MeshHierarchy
    multigrid_v_cycle(H,x0,b;nPre,nPost,alpha) src/solvers.jl:19-50  multigrid_v_cycle(...)
    ldiv!(H,b) / ldiv!(y,H,b)      src/solvers.jl:63-92   ldiv(H, b) / ldiv(y, H, b)
    multigrid(H,x0,b,maxiter,tol)  src/solvers.jl:116-139 multigrid(...)
    iterative_smoother_solve(A,S,x0,b;maxiter,tol,alpha)  iterative_smoother_solve(...)
                                   src/solvers.jl:189-213

Julia Symbols become strings.  Every compute step goes through libaggmg_hip.so (include/
aggmg_hip.h); nothing here computes the hot path on the CPU.  Mesh arguments only need the
fields the reference reads on this path: `mElements[i].mNodesInd` (1-based) and `mP`, or a
ready `mBlockInds` array (see uniform.py for the O(n) descriptors used at scale).
"""
import ctypes

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import _lib
from ._lib import (ArgumentError, DimensionMismatch, check)

_PD = ctypes.POINTER(ctypes.c_double)
_PI64 = ctypes.POINTER(ctypes.c_int64)


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _pd(a):
    return a.ctypes.data_as(_PD)


# --------------------------------------------------------------------------------------------
# context
# --------------------------------------------------------------------------------------------
class Context:
    """One HIP device + stream (aggmg_create).  Not thread-safe, like the reference."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        check(self.lib.aggmg_create(int(device), ctypes.byref(h)), None)
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.aggmg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, status):
        check(status, self.handle)

    def synchronize(self):
        self.check(self.lib.aggmg_synchronize(self.handle))

    def set_option(self, option, value):
        """context options of the C ABI (aggmg_set_option), e.g. _lib.OPT_SYMMETRIC_PACKING"""
        self.check(self.lib.aggmg_set_option(self.handle, int(option), int(value)))
        self.__dict__.setdefault("_options", {})[int(option)] = int(value)

    def option(self, option, default):
        """the value last given to set_option (the C ABI has no getter), `default` when it never was set"""
        return self.__dict__.get("_options", {}).get(int(option), default)

    def set_stream(self, hip_stream):
        """Launch on a caller-provided hipStream_t (integer / pointer), e.g.
        torch.cuda.current_stream().cuda_stream; 0 / None is the device's default stream."""
        self.check(self.lib.aggmg_set_stream(self.handle, ctypes.c_void_p(hip_stream or None)))

    def reset_stream(self):
        """back to the context's own non-blocking stream"""
        self.check(self.lib.aggmg_reset_stream(self.handle))

    # raw device vectors (harness plumbing; torch tensors' data_ptr() work equally well)
    def alloc(self, n):
        return DeviceVector(self, n)

    def to_device(self, x):
        x = _f64(x)
        v = DeviceVector(self, x.size)
        v.upload(x)
        return v

    def pin(self, x):
        """page-lock the memory of a NumPy array for good (C ABI aggmg_host_register): host-pointer calls that pass it
        -- multigrid_v_cycle(H, x0, b, out=), ldiv -- then move it by one DMA transfer instead of staging it.  The
        array must outlive the registration: unpin(x) (or freeing the context) ends it."""
        x = np.asarray(x)
        if x.dtype != np.float64 or not x.flags["C_CONTIGUOUS"]:
            raise ArgumentError("Context.pin: a C-contiguous float64 array")
        self.check(self.lib.aggmg_host_register(self.handle, ctypes.c_void_p(x.ctypes.data), x.nbytes))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[x.ctypes.data] = x      # (keeps the pages alive while they are registered)
        return x

    def unpin(self, x):
        self.check(self.lib.aggmg_host_unregister(self.handle, ctypes.c_void_p(x.ctypes.data)))
        getattr(self, "_pinned", {}).pop(x.ctypes.data, None)

    def pinned_empty(self, n):
        """float64 NumPy array of n entries in page-locked memory owned by the context (C ABI aggmg_host_alloc); freed
        with the context"""
        p = ctypes.c_void_p()
        self.check(self.lib.aggmg_host_alloc(self.handle, int(n) * 8, ctypes.byref(p)))
        buf = (ctypes.c_double * int(n)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.float64, count=int(n))

    def profile_enable(self, on=True):
        """True / 1: HIP events around every launch; 2: only the fine-level fused-down launch;
        False / 0: off"""
        self.check(self.lib.aggmg_profile_enable(self.handle, int(on)))

    def profile_collect(self):
        """-> {(kind_name, level): (total_ms, count)} for every tag seen since the last call."""
        ms = np.zeros(_lib.PROFILE_NTAGS)
        cnt = np.zeros(_lib.PROFILE_NTAGS, dtype=np.int64)
        self.check(self.lib.aggmg_profile_collect(self.handle, _pd(ms), cnt.ctypes.data_as(_PI64)))
        out = {}
        for tag in np.nonzero(cnt)[0]:
            out[(_lib.KIND_NAMES[tag // 16], int(tag % 16))] = (float(ms[tag]), int(cnt[tag]))
        return out


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class _DevPtr(ctypes.c_void_p):
    """Device address handed out by DeviceVector.ptr.  It keeps its vector referenced for as long as the pointer object
    itself lives -- the argument tuple of a ctypes call included -- so `ctx.to_device(x).ptr` passed straight into a
    `*_dev` entry point cannot be finalised (aggmg_dev_free) before the call has enqueued its launches; the free that
    follows synchronises the context's stream first (aggmg_dev_free), i.e. it waits for them.  (r03: such a temporary was
    freed between `.ptr` and the launch -- a dangling device pointer, a memory access fault on the GPU.)"""
    _owner = None


class DeviceVector:
    """fp64 vector in HBM owned by a Context.  `.ptr` is the device address (a c_void_p that keeps the vector alive);
    after free() it raises instead of yielding a stale address."""

    def __init__(self, ctx, n):
        self.ctx = ctx
        self.n = int(n)
        self._p = None
        p = ctypes.c_void_p()
        ctx.check(ctx.lib.aggmg_dev_alloc(ctx.handle, self.n * 8, ctypes.byref(p)))
        self._p = p

    @property
    def ptr(self):
        if self._p is None:
            raise ArgumentError("DeviceVector: used after free() -- its device memory has been released")
        q = _DevPtr(self._p.value)
        q._owner = self
        return q

    def upload(self, x):
        x = _f64(x)
        if x.size != self.n:
            raise DimensionMismatch("DeviceVector.upload: size mismatch")
        self.ctx.check(self.ctx.lib.aggmg_memcpy_h2d(self.ctx.handle, self.ptr, x.ctypes.data, x.size * 8))

    def download(self):
        out = np.empty(self.n)
        self.ctx.check(self.ctx.lib.aggmg_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.n * 8))
        return out

    def free(self):
        """release the device memory (waits for the launches already enqueued on the context's stream)"""
        p, self._p = self._p, None
        if p is not None and self.ctx.handle:
            self.ctx.lib.aggmg_dev_free(self.ctx.handle, p)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(v):
    """device pointer of a DeviceVector / torch tensor / int"""
    if v is None:
        return ctypes.c_void_p(None)
    if isinstance(v, DeviceVector):
        return v.ptr
    if hasattr(v, "data_ptr"):
        return ctypes.c_void_p(v.data_ptr())
    return ctypes.c_void_p(int(v))


# --------------------------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------------------------
class DeviceOperator:
    """Device mirror of one SparseMatrixCSC{Float64,Int64} (H.mStiffness[k] / H.mInterpolation[k],
    src/mesh_heirarchy.jl:20,26).  `A` is a SciPy sparse matrix or the Julia triple
    (m, n, colptr, rowval, nzval) with 1-based Int64 indices (a sixth entry 0 marks 0-based index arrays)."""

    def __init__(self, A, kind=_lib.OP_STIFFNESS, ctx=None):
        self.ctx = ctx or default_context()
        if isinstance(A, tuple):
            # (m, n, colptr, rowval, nzval[, one_based]): the arrays of a SparseMatrixCSC as they lie in memory
            m, n, colptr, rowval, nzval = A[:5]
            one_based = int(A[5]) if len(A) > 5 else 1
        else:
            if isinstance(A, np.ndarray) and A.ndim == 2:
                # a dense Matrix{Float64} in mInterpolation::Vector{AbstractMatrix{Float64}} (src/mesh_heirarchy.jl:26;
                # dg_cg_interpolation(..., 0) returns one): its non-zero entries as a CSC operator -- `L * v` and
                # `L' * v` add the same products in the same order
                A = sp.csc_matrix(A)
            elif not sp.issparse(A):
                raise ArgumentError("DeviceOperator: a SparseMatrixCSC (SciPy sparse matrix), the (m, n, colptr, rowval, "
                                    "nzval) arrays of one, or a dense 2-D array")
            A = sp.csc_matrix(A)
            A.sort_indices()
            m, n = A.shape
            colptr, rowval, nzval = A.indptr, A.indices, A.data
            one_based = 0
        colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        rowval = np.ascontiguousarray(rowval, dtype=np.int64)
        nzval = _f64(nzval)
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.aggmg_csc_upload(
            self.ctx.handle, int(m), int(n), colptr.ctypes.data_as(_PI64), rowval.ctypes.data_as(_PI64),
            _pd(nzval), one_based, int(kind), ctypes.byref(h)))
        self.handle = h
        self.shape = (int(m), int(n))
        self.nnz = int(nzval.size)
        self.kind = kind

    @classmethod
    def _from_handle(cls, ctx, handle, kind):
        """wrap an operator the library produced (aggmg_sp_matmul, aggmg_bd_sp_apply, ...)"""
        self = cls.__new__(cls)
        self.ctx, self.handle, self.kind = ctx, handle, kind
        m, n, nnz = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        ctx.check(ctx.lib.aggmg_op_shape(ctx.handle, handle, ctypes.byref(m), ctypes.byref(n), ctypes.byref(nnz)))
        self.shape, self.nnz = (m.value, n.value), nnz.value
        return self

    def to_scipy(self):
        """the operator as a SciPy CSC matrix (C ABI aggmg_op_download_csc): stored entries, zeros included"""
        cp = np.empty(self.shape[1] + 1, dtype=np.int32)
        rv = np.empty(self.nnz, dtype=np.int32)
        nz = np.empty(self.nnz)
        self.ctx.check(self.ctx.lib.aggmg_op_download_csc(
            self.ctx.handle, self.handle, cp.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
            rv.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _pd(nz)))
        return sp.csc_matrix((nz, rv, cp), shape=self.shape)

    def _binary(self, fn, other, kind):
        other = _as_op(other, kind, self.ctx)
        h = ctypes.c_void_p()
        self.ctx.check(fn(self.ctx.handle, self.handle, other.handle, int(kind), ctypes.byref(h)))
        return DeviceOperator._from_handle(self.ctx, h, kind)

    def matmul(self, B, kind=_lib.OP_STIFFNESS):
        """`A * B` for two sparse matrices on the device (C ABI aggmg_sp_matmul): the products of
        `L'*X*L` and `D*(M_LU\\G)`, src/mesh_heirarchy.jl:71-72,79-84"""
        return self._binary(self.ctx.lib.aggmg_sp_matmul, B, kind)

    def sub(self, B, kind=_lib.OP_STIFFNESS):
        """`A - B`, numerically-zero results dropped as SparseArrays does (C ABI aggmg_sp_sub)"""
        return self._binary(self.ctx.lib.aggmg_sp_sub, B, kind)

    def transpose(self, kind=_lib.OP_STIFFNESS):
        """the adjoint as an operator of its own (C ABI aggmg_op_transpose)"""
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.aggmg_op_transpose(self.ctx.handle, self.handle, int(kind), ctypes.byref(h)))
        return DeviceOperator._from_handle(self.ctx, h, kind)

    def download(self, transposed=False):
        """Device index maps and values: (rowptr, colind, vals), 0-based int32 CSR of the matrix
        (or of its transpose)."""
        nrows = self.shape[1] if transposed else self.shape[0]
        rp = np.empty(nrows + 1, dtype=np.int32)
        ci = np.empty(self.nnz, dtype=np.int32)
        v = np.empty(self.nnz)
        self.ctx.check(self.ctx.lib.aggmg_op_download(
            self.ctx.handle, self.handle, 1 if transposed else 0,
            rp.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
            ci.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _pd(v)))
        return rp, ci, v

    def release_host(self):
        self.ctx.check(self.ctx.lib.aggmg_op_release_host(self.ctx.handle, self.handle))

    def free(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx.lib.aggmg_op_free(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _as_op(A, kind=_lib.OP_STIFFNESS, ctx=None):
    return A if isinstance(A, DeviceOperator) else DeviceOperator(A, kind, ctx)


# --------------------------------------------------------------------------------------------
# smoothers (src/smoother.jl)
# --------------------------------------------------------------------------------------------
class AbstractSmoother:
    """abstract type AbstractSmoother (src/AgglomerationMultigrid1D.jl:16)"""
    handle = None

    def free(self):
        if getattr(self, "handle", None) and self.A.ctx.handle:
            self.A.ctx.lib.aggmg_smoother_free(self.A.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def structured(self):
        out = ctypes.c_int(0)
        c = self.A.ctx
        c.check(c.lib.aggmg_smoother_is_structured(c.handle, self.handle, ctypes.byref(out)))
        return bool(out.value)


class JacobiSmoother(AbstractSmoother):
    """JacobiSmoother{mJac::Diagonal} src/smoother.jl:52-58.

    mElementNodes: optional (p+1) x n matrix of the CG mesh's element node lists (1-based, column k
    = cgMesh.mElements[k].mNodesInd, src/cg_mesh.jl:35-45) -- what cg_smoother(cgMesh, A, :jac) has at
    hand.  Same smoother; with the lists the level also gets the element-contiguous chain form and
    runs the fused point-Jacobi kernel (C ABI aggmg_jacobi_setup_elements).
    Without lists the library looks at the operator itself (AGGMG_OPT_DETECT_CHAIN): a CG operator in the reference's
    vertices-first numbering is recognised from its pattern and takes the same chain kernels; detect=False keeps an
    operator without lists on the generic CSR kernels whatever its pattern."""

    def __init__(self, A, ctx=None, mElementNodes=None, detect=True):
        self.A = _as_op(A, ctx=ctx)
        h = ctypes.c_void_p()
        c = self.A.ctx
        if mElementNodes is None:
            was = c.option(_lib.OPT_DETECT_CHAIN, 1)
            if not detect and was:
                c.set_option(_lib.OPT_DETECT_CHAIN, 0)
            try:
                c.check(c.lib.aggmg_jacobi_setup(c.handle, self.A.handle, ctypes.byref(h)))
            finally:
                if not detect and was:
                    c.set_option(_lib.OPT_DETECT_CHAIN, was)
        else:
            inds = np.asarray(mElementNodes, dtype=np.int64)
            if inds.ndim != 2:
                raise ArgumentError("mElementNodes must be a ((p+1) x n) matrix")
            m1, n = inds.shape
            flat = np.ascontiguousarray(inds.T)     # column-major (p+1) x n == C-order n x (p+1)
            c.check(c.lib.aggmg_jacobi_setup_elements(c.handle, self.A.handle, m1, n, flat.ctypes.data_as(_PI64), 1,
                                                      ctypes.byref(h)))
        self.handle = h


class _BlockSmoother(AbstractSmoother):
    _kind = 0

    def __init__(self, A, mBlockInds, ctx=None):
        """mBlockInds: Matrix{Int64}(m x nb), 1-based, column i = index list of block i
        (src/smoother.jl:13,76)."""
        self.A = _as_op(A, ctx=ctx)
        inds = np.asarray(mBlockInds, dtype=np.int64)
        if inds.ndim != 2:
            raise ArgumentError("mBlockInds must be an (m x nb) matrix")
        self.mBlockInds = inds
        m, nb = inds.shape
        flat = np.ascontiguousarray(inds.T)  # column-major m x nb == C-order nb x m
        h = ctypes.c_void_p()
        c = self.A.ctx
        c.check(c.lib.aggmg_blockjacobi_setup(c.handle, self.A.handle, m, nb,
                                              flat.ctypes.data_as(_PI64), 1, self._kind,
                                              ctypes.byref(h)))
        self.handle = h

    def inverse_blocks(self):
        """the explicit inverses of the pivoted block LUs, (nb, m, m): what the device factorisation (K6)
        produced from A[inds, inds] (C ABI aggmg_smoother_download_blocks)"""
        m, nb = self.mBlockInds.shape
        return _download_blocks(self.A.ctx, self.handle, nb, m)


def _download_blocks(ctx, handle, nb, m):
    out = np.empty((nb, m, m))
    ctx.check(ctx.lib.aggmg_smoother_download_blocks(ctx.handle, handle, _pd(out)))
    return out


class BlockJacobi(_BlockSmoother):
    """BlockJacobi{mBlocks::Vector{LU}, mBlockInds} src/smoother.jl:64-81.  Blocks are
    re-extracted from A on set-up (identical mathematics to src/smoother.jl:159-162)."""


class AdditiveSchwarzSmoother(_BlockSmoother):
    """src/smoother.jl:1-18 (same apply loop as BlockJacobi, overlapping blocks)"""


class BlockGaussSeidel(_BlockSmoother):
    """EXTENSION: red-black block Gauss-Seidel on a block-tridiagonal operator (C ABI kind 2 of
    aggmg_blockjacobi_setup).  The reference has no Gauss-Seidel smoother;
    BASELINE.json names one, so it is offered and checked against the oracle's restatement
    (BlockGaussSeidelRB), not against the reference.  Also on the overlapping element blocks of a CG mesh
    (cg_smoother(mesh, A, 'blockGS'): elements of one colour share no node).  One sweep: even elements, then odd ones, each
    with the newest values; V-cycles post-smooth in the reverse order.  UnsupportedError unless the
    blocks are contiguous and the operator couples an element to its direct neighbours only."""
    _kind = 2


class HybridSchwarzSmoother(_BlockSmoother):
    """src/smoother.jl:24-46"""
    _kind = 1


class _BlockObject(AbstractSmoother):
    """device handle of a block-diagonal matrix or of its factorisation"""

    def __init__(self, ctx, handle, N):
        self._ctx, self.handle, self.N = ctx, handle, N

    def free(self):
        if getattr(self, "handle", None) and self._ctx.handle:
            self._ctx.lib.aggmg_smoother_free(self._ctx.handle, self.handle)
        self.handle = None

    def _apply_sparse(self, S, kind):
        """block object times a sparse matrix -> DeviceOperator (C ABI aggmg_bd_sp_apply)"""
        S = _as_op(S, kind, self._ctx)
        h = ctypes.c_void_p()
        c = self._ctx
        c.check(c.lib.aggmg_bd_sp_apply(c.handle, self.handle, S.handle, int(kind), ctypes.byref(h)))
        return DeviceOperator._from_handle(c, h, kind)

    def _apply(self, B):
        B = np.asarray(B, dtype=np.float64)
        if B.ndim not in (1, 2) or B.shape[0] != self.N:
            raise DimensionMismatch("BlockDiagonal: DimensionMismatch")
        ncols = 1 if B.ndim == 1 else B.shape[1]
        Bf = np.asfortranarray(B.reshape(self.N, ncols))
        Y = np.empty((self.N, ncols), order='F')
        c = self._ctx
        c.check(c.lib.aggmg_smoother_apply(c.handle, self.handle, Bf.ctypes.data_as(_PD), self.N, ncols, 1.0,
                                           Y.ctypes.data_as(_PD)))
        return Y[:, 0].copy() if B.ndim == 1 else np.ascontiguousarray(Y)


class BlockDiagonal:
    """struct BlockDiagonal{mBlocks, mBlockSize, mBlockInds} (src/block_diagonal.jl:11-15) with the
    BlockDiagonal(mBlocks) constructor (:27-41): equal-sized dense blocks, contiguous index lists.
    `A * x` / `A * B` for dense vectors and matrices run the batched block kernel (mul!, :166-176);
    `lu(A)` (:276) gives a BlockDiagonalLU."""

    def __init__(self, mBlocks, ctx=None):
        if len(mBlocks) == 0:
            raise ArgumentError("BlockDiagonal needs at least one block")
        if isinstance(mBlocks, np.ndarray) and mBlocks.ndim == 3:
            # the blocks as one (nb, m, m) array (what the vectorised builders produce): no per-block Python work
            if mBlocks.shape[1] != mBlocks.shape[2]:
                raise ArgumentError("All blocks must be of the same size.")
            m = mBlocks.shape[1]
            self.mBlocks = np.ascontiguousarray(mBlocks, dtype=np.float64)
        else:
            m = np.asarray(mBlocks[0]).shape[0]
            for blk in mBlocks:
                if np.asarray(blk).shape != (m, m):
                    raise ArgumentError("All blocks must be of the same size.")   # block_diagonal.jl:35-37
            self.mBlocks = [np.array(blk, dtype=np.float64) for blk in mBlocks]
        self.mBlockSize = m
        nb = len(mBlocks)
        self.mBlockInds = np.arange(nb, dtype=np.int64)[None, :] * m + np.arange(1, m + 1, dtype=np.int64)[:, None]
        self._ctx = ctx or default_context()
        self._dev = None

    @property
    def shape(self):
        n = len(self.mBlocks) * self.mBlockSize
        return (n, n)

    def _setup(self, factorize):
        c = self._ctx
        if isinstance(self.mBlocks, np.ndarray):
            flat = np.ascontiguousarray(np.transpose(self.mBlocks, (0, 2, 1)))
        else:
            flat = np.ascontiguousarray(np.stack([b.T for b in self.mBlocks]))   # column-major per block
        h = ctypes.c_void_p()
        c.check(c.lib.aggmg_blockdiag_setup(c.handle, self.mBlockSize, len(self.mBlocks), flat.ctypes.data_as(_PD),
                                            1 if factorize else 0, ctypes.byref(h)))
        return _BlockObject(c, h, self.shape[0])

    def __matmul__(self, B):
        """`A * x`, `A * B` (dense: mul!, :166-176) and `A * S` for a sparse S (SciPy sparse or DeviceOperator:
        bd_sp_matmul, :195-264 -> DeviceOperator)"""
        if self._dev is None:
            self._dev = self._setup(False)
        if sp.issparse(B) or isinstance(B, DeviceOperator):
            return self._dev._apply_sparse(B, _lib.OP_STIFFNESS)
        return self._dev._apply(B)

    mul = __matmul__

    def lu(self):
        return BlockDiagonalLU(self)

    def todense(self):
        """Matrix(A) (src/block_diagonal.jl:107-114)"""
        out = np.zeros(self.shape)
        m = self.mBlockSize
        for k, blk in enumerate(self.mBlocks):
            out[k * m:(k + 1) * m, k * m:(k + 1) * m] = blk
        return out


class BlockDiagonalLU:
    """struct BlockDiagonalLU (src/block_diagonal.jl:17-21, ctor :47-58): `A \\ x`, `A \\ B` for dense
    right-hand sides (ldiv!, :299-309).  A singular block raises SingularException."""

    def __init__(self, A):
        self.mBlockSize, self.mBlockInds = A.mBlockSize, A.mBlockInds
        self._dev = A._setup(True)

    @property
    def shape(self):
        return (self._dev.N, self._dev.N)

    def solve(self, B, kind=_lib.OP_STIFFNESS):
        """`A \\ x`, `A \\ B` (dense: ldiv!, :299-309) and `A \\ S` for a sparse S (bd_sp_solve, :314-383
        -> DeviceOperator).  Deviation from the reference's operation sequence: every block's pivoted LU is turned
        into an explicit inverse once and applied by multiplication (the reference runs getrs per block and column,
        src/block_diagonal.jl:305,374): equal to round-off for the well-conditioned mass blocks of the hierarchy
        constructors, checked against the oracle at 1e-12."""
        if sp.issparse(B) or isinstance(B, DeviceOperator):
            return self._dev._apply_sparse(B, kind)
        return self._dev._apply(B)

    ldiv = solve


def _mesh_block_inds(mesh):
    if hasattr(mesh, "mBlockInds"):
        return np.asarray(mesh.mBlockInds, dtype=np.int64)
    n = len(mesh.mElements)
    inds = np.zeros((mesh.mP + 1, n), dtype=np.int64)
    for i, el in enumerate(mesh.mElements):
        inds[:, i] = el.mNodesInd
    return inds


def dg_smoother(dgMesh, A, smootherType, ctx=None):
    """dg_smoother(mesh, A, :jac | :blockJac) src/smoother.jl:142-168"""
    if smootherType == 'jac':
        return JacobiSmoother(A, ctx)
    if smootherType == 'blockJac':
        return BlockJacobi(A, _mesh_block_inds(dgMesh), ctx)
    if smootherType == 'blockGS':    # extension, see BlockGaussSeidel
        return BlockGaussSeidel(A, _mesh_block_inds(dgMesh), ctx)
    raise ArgumentError(f"dg_smoother: unknown smoother type {smootherType!r}")


def cg_smoother(cgMesh, A, smootherType, ctx=None):
    """cg_smoother(mesh, A, :jac | :addSchwarz | :hybridSchwarz) src/smoother.jl:88-139"""
    if smootherType == 'jac':
        has_elements = cgMesh is not None and (hasattr(cgMesh, "mBlockInds") or hasattr(cgMesh, "mElements"))
        return JacobiSmoother(A, ctx, _mesh_block_inds(cgMesh) if has_elements else None)
    if smootherType == 'addSchwarz':
        return AdditiveSchwarzSmoother(A, _mesh_block_inds(cgMesh), ctx)
    if smootherType == 'hybridSchwarz':
        return HybridSchwarzSmoother(A, _mesh_block_inds(cgMesh), ctx)
    if smootherType == 'blockGS':    # extension: red-black ELEMENT Gauss-Seidel on the element chain, see BlockGaussSeidel
        return BlockGaussSeidel(A, _mesh_block_inds(cgMesh), ctx)
    raise ArgumentError(f"cg_smoother: unknown smoother type {smootherType!r}")


def apply_smoother(S, B, alpha=1.0):
    """apply_smoother(A::AbstractSmoother, B::AbstractVecOrMat; alpha=1.0) -> new array
    (src/smoother.jl:6,30,56,69).  B is not modified."""
    B = np.asarray(B, dtype=np.float64)
    if B.ndim not in (1, 2):
        raise DimensionMismatch("apply_smoother: B must be a vector or a matrix")
    N = B.shape[0]
    ncols = 1 if B.ndim == 1 else B.shape[1]
    Bf = np.asfortranarray(B.reshape(N, ncols))
    Y = np.empty((N, ncols), order='F')
    c = S.A.ctx
    c.check(c.lib.aggmg_smoother_apply(c.handle, S.handle, Bf.ctypes.data_as(_PD), N, ncols,
                                       float(alpha), Y.ctypes.data_as(_PD)))
    return Y[:, 0].copy() if B.ndim == 1 else np.ascontiguousarray(Y)


# --------------------------------------------------------------------------------------------
# hierarchy (src/mesh_heirarchy.jl) and solvers (src/solvers.jl)
# --------------------------------------------------------------------------------------------
def _is_cg_mesh(mesh):
    """a CgMesh shares vertices between elements (src/cg_mesh.jl:37-45): n p + 1 nodes on n elements"""
    try:
        els = mesh.mElements
        return mesh.mNumNodes == len(els) * mesh.mP + 1 and mesh.mP >= 1
    except AttributeError:
        return False


def _smoother_from_reference(S, A_op, mesh=None):
    """Accept the reference's smoother objects (anything with `mJac`, or `mBlockInds` [+
    `mCountingMatrix`]) as well as this module's; blocks are re-extracted from A on device.  A
    JacobiSmoother on a CG mesh also receives the mesh's element node lists (what
    cg_smoother(cgMesh, A, :jac) has at hand, src/smoother.jl:88-102)."""
    if isinstance(S, AbstractSmoother):
        return S
    if hasattr(S, "mCountingMatrix"):
        return HybridSchwarzSmoother(A_op, S.mBlockInds)
    if type(S).__name__ == "BlockGaussSeidelRB":     # the oracle's restatement of the extension
        return BlockGaussSeidel(A_op, S.mBlockInds)
    if hasattr(S, "mBlockInds"):
        return BlockJacobi(A_op, S.mBlockInds)
    if hasattr(S, "mJac"):
        if mesh is not None and _is_cg_mesh(mesh):
            return JacobiSmoother(A_op, None, _mesh_block_inds(mesh))
        return JacobiSmoother(A_op)
    raise ArgumentError("unrecognised smoother object")


class MeshHierarchy:
    """struct MeshHierarchy (src/mesh_heirarchy.jl:17-28) with device mirrors of the operator
    vectors.  Fields keep the reference's names; level 1 (index 0) is the finest.

    MeshHierarchy(mMeshes, mStiffness, mSmoothers, mInterpolation, mBdConds=None, ...) is the
    struct's positional constructor (the reference's two outer constructors do Galerkin set-up,
    which is outside the hot path: SURVEY.md 8 a12/a13); `from_reference(H)` wraps any object
    carrying the reference's fields (e.g. the set-up output of a Julia-side or oracle-side
    constructor)."""

    def __init__(self, mMeshes, mStiffness, mSmoothers, mInterpolation, mBdConds=None,
                 mGradient=None, mDivergence=None, mC=None, ctx=None, keep_host=True,
                 coarse_mode=_lib.COARSE_AUTO):
        n = len(mStiffness)
        if n < 1:
            raise ArgumentError("At least one mesh required.")
        if mMeshes is not None and len(mMeshes) != n:
            raise ArgumentError("Length of vector of meshes does not match the operators.")
        if len(mInterpolation) != n - 1 or len(mSmoothers) < n - 1:
            raise ArgumentError("Need one interpolation and one smoother per non-coarsest level.")
        self.ctx = ctx or default_context()
        self.mMeshes = mMeshes
        self.mStiffness = list(mStiffness)
        self.mInterpolation = list(mInterpolation)
        self.mBdConds = mBdConds
        self.mGradient, self.mDivergence, self.mC = mGradient, mDivergence, mC
        self._ops = [_as_op(A, _lib.OP_STIFFNESS, self.ctx) for A in mStiffness]
        self._Ls = [_as_op(L, _lib.OP_TRANSFER, self.ctx) for L in mInterpolation]
        self.mSmoothers = [_smoother_from_reference(mSmoothers[k], self._ops[k],
                                                    mMeshes[k] if mMeshes is not None else None)
                           for k in range(n - 1)]
        arr = ctypes.c_void_p * n
        ops = arr(*[o.handle for o in self._ops])
        sms = arr(*([s.handle for s in self.mSmoothers] + [None]))
        Ls = arr(*([l.handle for l in self._Ls] + [None]))
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.aggmg_hier_create(self.ctx.handle, n, ops, sms, Ls, int(coarse_mode), ctypes.byref(h)))
        self.handle = h
        if not keep_host:
            for o in self._ops + self._Ls:
                o.release_host()

    @classmethod
    def from_reference(cls, H, ctx=None, coarse_mode=_lib.COARSE_AUTO):
        return cls(H.mMeshes, H.mStiffness, H.mSmoothers, H.mInterpolation,
                   getattr(H, "mBdConds", None), getattr(H, "mGradient", None),
                   getattr(H, "mDivergence", None), getattr(H, "mC", None), ctx=ctx,
                   coarse_mode=coarse_mode)

    @classmethod
    def from_dg_operators(cls, mMeshes, A, G, D, C, mInterpolation, mMassMatrices, ctx=None, keep_host=True,
                          coarse_mode=_lib.COARSE_AUTO):
        """MeshHierarchy(mMeshes, mBdConds, A, G, D, C; nDG, nAgg) (src/mesh_heirarchy.jl:140-181, with the
        agglomerated levels of SURVEY D4) given the interpolation matrices L_k and the mass matrices
        (BlockDiagonal) of the coarser levels: the recurrences of :79-84 / :98-103 run on the device --

            G_{k+1} = L' * G_k * L,  D_{k+1} = L' * D_k * L,  C_{k+1} = L' * C_k * L
            A_{k+1} = C_{k+1} - D_{k+1} * (M_LU \\ G_{k+1})
            mSmoothers[k] = dg_smoother(mMeshes[k], A_k, :blockJac)

        (aggmg_op_transpose, aggmg_sp_matmul, aggmg_bd_sp_apply, aggmg_sp_sub, aggmg_blockjacobi_setup).
        mMeshes[k] need `mBlockInds` (or `mElements` / `mP`); mMassMatrices[k] is the BlockDiagonal of
        level k + 1.  The Galerkin operators stay in HBM; H.mGradient / mDivergence / mC / mStiffness hold
        DeviceOperators (`.to_scipy()` reads one back)."""
        ctx = ctx or default_context()
        n = len(mMeshes)
        if n < 1:
            raise ArgumentError("At least one DG mesh required.")
        if len(mInterpolation) != n - 1 or len(mMassMatrices) != n - 1:
            raise ArgumentError("Length of vector of meshes does not match inputed number of DG and "
                                "agglomerated meshes.")
        St = [_as_op(A, _lib.OP_STIFFNESS, ctx)]
        Gs, Ds, Cs = [_as_op(G, _lib.OP_STIFFNESS, ctx)], [_as_op(D, _lib.OP_STIFFNESS, ctx)], [_as_op(C, _lib.OP_STIFFNESS, ctx)]
        Ls = [_as_op(L, _lib.OP_TRANSFER, ctx) for L in mInterpolation]
        for k in range(n - 1):
            L = Ls[k]
            Lt = L.transpose()
            Gs.append(Lt.matmul(Gs[k]).matmul(L))         # (L' * G) * L, left to right as Julia evaluates it
            Ds.append(Lt.matmul(Ds[k]).matmul(L))
            Cs.append(Lt.matmul(Cs[k]).matmul(L))
            M = mMassMatrices[k]
            Mlu = M if isinstance(M, BlockDiagonalLU) else M.lu()
            St.append(Cs[k + 1].sub(Ds[k + 1].matmul(Mlu.solve(Gs[k + 1]))))
            Lt.free()
        sms = [BlockJacobi(St[k], _mesh_block_inds(mMeshes[k]), ctx) for k in range(n - 1)]
        H = cls(mMeshes, St, sms, Ls, mGradient=Gs, mDivergence=Ds, mC=Cs, ctx=ctx, keep_host=keep_host,
                coarse_mode=coarse_mode)
        return H

    @classmethod
    def from_cg_operators(cls, mMeshes, A, mInterpolation, nCG, dg_operators=None, mMassMatrices=(), ctx=None,
                          keep_host=True, coarse_mode=_lib.COARSE_AUTO):
        """MeshHierarchy(mMeshes, mesh, mBdConds, A; nCG, nDG, nAgg, CDir) (src/mesh_heirarchy.jl:30-138) given
        the interpolation matrices of all levels: the CG chain's Galerkin operators `L'*A*L` (:53-59) and, below
        the first DG / agglomerated level, the recurrences of :76-86 / :89-106 run on the device.  That first
        DG level is RE-DISCRETISED in the reference (:65-72, dg_flux_operators -- assembly, outside the hot path):
        its G, D, C come in through dg_operators = (G, D, C); mMassMatrices[i] is the BlockDiagonal mass matrix
        of level nCG + i.  CG levels get cg_smoother(mesh, A, :jac) with the mesh's element lists, DG levels
        dg_smoother(mesh, A, :blockJac)."""
        ctx = ctx or default_context()
        n = len(mMeshes)
        if nCG <= 0:
            raise ArgumentError("At least one CG mesh required.")
        if len(mInterpolation) != n - 1 or (n > nCG and (dg_operators is None or len(mMassMatrices) != n - nCG)):
            raise ArgumentError("Length of vector of meshes does not match inputed number of CG, DG, and "
                                "agglomerated meshes.")
        Ls = [_as_op(L, _lib.OP_TRANSFER, ctx) for L in mInterpolation]
        St = [_as_op(A, _lib.OP_STIFFNESS, ctx)]
        for i in range(nCG - 1):
            Lt = Ls[i].transpose()
            St.append(Lt.matmul(St[i]).matmul(Ls[i]))
            Lt.free()
        Gs, Ds, Cs = [], [], []
        if n > nCG:
            Gs.append(_as_op(dg_operators[0], _lib.OP_STIFFNESS, ctx))
            Ds.append(_as_op(dg_operators[1], _lib.OP_STIFFNESS, ctx))
            Cs.append(_as_op(dg_operators[2], _lib.OP_STIFFNESS, ctx))
            lus = [M if isinstance(M, BlockDiagonalLU) else M.lu() for M in mMassMatrices]
            St.append(Cs[0].sub(Ds[0].matmul(lus[0].solve(Gs[0]))))
            for i in range(1, n - nCG):
                L = Ls[nCG + i - 1]
                Lt = L.transpose()
                Gs.append(Lt.matmul(Gs[i - 1]).matmul(L))
                Ds.append(Lt.matmul(Ds[i - 1]).matmul(L))
                Cs.append(Lt.matmul(Cs[i - 1]).matmul(L))
                St.append(Cs[i].sub(Ds[i].matmul(lus[i].solve(Gs[i]))))
                Lt.free()
        sms = []
        for k in range(n - 1):
            if k < nCG:
                sms.append(cg_smoother(mMeshes[k], St[k], 'jac', ctx))
            else:
                sms.append(BlockJacobi(St[k], _mesh_block_inds(mMeshes[k]), ctx))
        return cls(mMeshes, St, sms, Ls, mGradient=Gs, mDivergence=Ds, mC=Cs, ctx=ctx, keep_host=keep_host,
                   coarse_mode=coarse_mode)

    @property
    def nlevels(self):
        return len(self._ops)

    def structured_levels(self):
        return [s.structured for s in self.mSmoothers]

    def level_kinds(self):
        """which kernels run each level: 'fused_btd', 'fused_chain', 'generic', 'coarsest' (C ABI
        aggmg_hier_level_kind)"""
        out = []
        for k in range(self.nlevels):
            v = ctypes.c_int(0)
            self.ctx.check(self.ctx.lib.aggmg_hier_level_kind(self.ctx.handle, self.handle, k, ctypes.byref(v)))
            out.append(_lib.LEVEL_KIND_NAMES[v.value])
        return out

    def paired_levels(self, nsweeps=3, direction="down"):
        """levels k whose launch also carries level k + 1 (C ABI aggmg_hier_level_paired), as the descent walks the
        hierarchy (pairs taken from the fine side) or, direction='up', as the ascent does (from the coarse side)"""
        def ok(k):
            v = ctypes.c_int(0)
            self.ctx.check(self.ctx.lib.aggmg_hier_level_paired(self.ctx.handle, self.handle, k, int(nsweeps), ctypes.byref(v)))
            return bool(v.value)
        out = []
        if direction == "down":
            k = 0
            while k < self.nlevels - 1:
                if ok(k):
                    out.append(k)
                    k += 1
                k += 1
        else:
            k = self.nlevels - 2
            while k >= 1:
                if ok(k - 1):
                    out.append(k - 1)
                    k -= 1
                k -= 1
        return sorted(out)

    def launch_bytes(self, level, kind, has_x0=None):
        """(read, write) compulsory HBM bytes of the fused launch of `level`: kind 'down' / 'up' / 'mid'
        (C ABI aggmg_hier_launch_bytes) -- every array of the launch counted once, as stored."""
        kk = {"down": _lib.KIND_FUSED_DOWN, "up": _lib.KIND_FUSED_UP, "mid": _lib.KIND_FUSED_MID}[kind]
        r, w = ctypes.c_int64(0), ctypes.c_int64(0)
        x0 = (level == 0) if has_x0 is None else bool(has_x0)
        self.ctx.check(self.ctx.lib.aggmg_hier_launch_bytes(self.ctx.handle, self.handle, int(level), kk, int(x0),
                                                            ctypes.byref(r), ctypes.byref(w)))
        return r.value, w.value

    def vcycle_dev(self, x0, b, x_out, nPre=3, nPost=3, alpha=2.0 / 3.0):
        """Device-resident V-cycle: x0, b, x_out are DeviceVector / torch tensors / raw pointers;
        asynchronous on the context stream."""
        c = self.ctx
        c.check(c.lib.aggmg_vcycle_dev(c.handle, self.handle, _ptr(x0), _ptr(b), int(nPre), int(nPost),
                                       float(alpha), _ptr(x_out)))

    def vcycles_dev(self, x0, b, x_out, ncycles, nPre=3, nPost=3, alpha=2.0 / 3.0):
        """ncycles device-resident V-cycles back to back (x <- V(x, b)), the loop body of
        multigrid (src/solvers.jl:124-126); between cycles the fine level's post- and pre-smoothing
        share one fused launch."""
        c = self.ctx
        c.check(c.lib.aggmg_vcycles_dev(c.handle, self.handle, _ptr(x0), _ptr(b), int(ncycles), int(nPre),
                                        int(nPost), float(alpha), _ptr(x_out)))

    def vcycle_down_dev(self, x0, b, nPre=3, alpha=2.0 / 3.0):
        """Descending half (src/solvers.jl:28-37); leaves the coarsest rhs in coarse_buffers()."""
        c = self.ctx
        c.check(c.lib.aggmg_vcycle_down_dev(c.handle, self.handle, _ptr(x0), _ptr(b), int(nPre), float(alpha)))

    def vcycle_up_dev(self, b, x_out, nPost=3, alpha=2.0 / 3.0):
        """Ascending half (src/solvers.jl:41-47); the coarsest solution must be in coarse_buffers()."""
        c = self.ctx
        c.check(c.lib.aggmg_vcycle_up_dev(c.handle, self.handle, _ptr(b), int(nPost), float(alpha), _ptr(x_out)))

    def vcycle_up_split_dev(self, b, x_out, head_elems, tail_elem, part, nPost=3, alpha=2.0 / 3.0):
        """the ascent in three calls (C ABI aggmg_vcycle_up_split_dev): part 0 = coarser levels, part 1 =
        the fine-level tiles holding elements [0, head_elems) and [tail_elem, ne), part 2 = the rest"""
        c = self.ctx
        c.check(c.lib.aggmg_vcycle_up_split_dev(c.handle, self.handle, _ptr(b), int(nPost), float(alpha), _ptr(x_out),
                                                int(head_elems), int(tail_elem), int(part)))

    def set_restriction(self, mode):
        """_lib.RESTRICT_EXPLICIT (default: the reference's arithmetic for L'(rhs - A u)) or
        _lib.RESTRICT_PRECONDITIONED (cheaper; UnsupportedError above
        _lib.RESTRICT_PRECONDITIONED_MAX_ELEMS fine elements, where its rounding error on the smoothest
        mode would stall or reverse convergence) -- C ABI aggmg_hier_set_restriction, include/aggmg_hip.h"""
        c = self.ctx
        c.check(c.lib.aggmg_hier_set_restriction(c.handle, self.handle, int(mode)))

    def get_restriction(self):
        v = ctypes.c_int(0)
        self.ctx.check(self.ctx.lib.aggmg_hier_get_restriction(self.ctx.handle, self.handle, ctypes.byref(v)))
        return v.value

    def coarse_buffers(self):
        """-> (rhs_ptr, sol_ptr, n): device buffers of the coarsest level"""
        r, s_, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64(0)
        self.ctx.check(self.ctx.lib.aggmg_hier_coarse_buffers(self.ctx.handle, self.handle, ctypes.byref(r),
                                                              ctypes.byref(s_), ctypes.byref(n)))
        return r.value, s_.value, n.value

    def coarse_info(self):
        """-> dict(on_device, block_size, cond_est, probe_backward_error, tail, tail_blocks) of the coarsest direct solver;
        probe_backward_error: ||d - A x|| / ||d|| of the probe solve that aggmg_hier_create accepts (< 1e-10) or rejects the
        device factorisation on (-1: none attempted)"""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_double(0.0)
        self.ctx.check(self.ctx.lib.aggmg_hier_coarse_info(self.ctx.handle, self.handle, ctypes.byref(a),
                                                           ctypes.byref(b), ctypes.byref(c)))
        e = ctypes.c_double(0.0)
        self.ctx.check(self.ctx.lib.aggmg_hier_coarse_probe(self.ctx.handle, self.handle, ctypes.byref(e)))
        k, nb = ctypes.c_int(0), ctypes.c_int64(0)
        self.ctx.check(self.ctx.lib.aggmg_hier_coarse_tail(self.ctx.handle, self.handle, ctypes.byref(k), ctypes.byref(nb)))
        return dict(on_device=bool(a.value), block_size=b.value, cond_est=c.value, probe_backward_error=e.value,
                    tail=("none", "cyclic reduction", "parallel cyclic reduction")[k.value], tail_blocks=nb.value)

    def last_coarse_ms(self):
        ms = ctypes.c_double(0.0)
        self.ctx.check(self.ctx.lib.aggmg_hier_last_coarse_ms(self.ctx.handle, self.handle, ctypes.byref(ms)))
        return ms.value

    def free(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx.lib.aggmg_hier_free(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def multigrid_v_cycle(H, x0, b, nPre=3, nPost=3, alpha=2.0 / 3.0, out=None):
    """multigrid_v_cycle(H, x0, b; nPre=3, nPost=3, alpha=2/3) -> x   (src/solvers.jl:19-50).
    Returns a new vector; x0 and b are not modified.  x0, b: host arrays (-> NumPy array) or DeviceVectors
    (-> DeviceVector, nothing leaves the device).  out (host arrays only, an extension): the array the result goes to
    instead of a new one -- with x0, b, out page-locked once (Context.pin / pinned_empty) a call moves its three vectors
    by DMA instead of staging them."""
    if not isinstance(nPre, (int, np.integer)) or not isinstance(nPost, (int, np.integer)):
        raise TypeError("nPre / nPost must be integers (nPre::Integer, src/solvers.jl:20)")
    if x0 is None and isinstance(b, DeviceVector):      # zero initial guess (ldiv!): nothing is read for it
        out = H.ctx.alloc(b.n)
        H.vcycle_dev(None, b, out, int(nPre), int(nPost), float(alpha))
        return out
    if isinstance(x0, DeviceVector) and isinstance(b, DeviceVector):
        # vectors already in HBM: nothing crosses PCIe, a new DeviceVector comes back (aggmg_vcycle_dev) -- the form
        # for callers that loop; host arrays cost 3 * 8 * N bytes of PCIe per call (bench.py: pcie_inclusive)
        N = H._ops[0].shape[0]
        if x0.n != N or b.n != N:
            raise DimensionMismatch("multigrid_v_cycle: x0 / b do not match the fine operator")
        out = H.ctx.alloc(N)
        H.vcycle_dev(x0, b, out, int(nPre), int(nPost), float(alpha))
        return out
    x0 = None if x0 is None else _f64(x0)     # None: a zero initial guess (what ldiv! uses), no vector of zeros sent
    b = _f64(b)
    N = H._ops[0].shape[0]
    if (x0 is not None and x0.shape != (N,)) or b.shape != (N,):
        raise DimensionMismatch("multigrid_v_cycle: x0 / b do not match the fine operator")
    if out is None:
        out = np.empty(N)
    elif not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == (N,) and out.flags["C_CONTIGUOUS"]):
        raise DimensionMismatch("multigrid_v_cycle: out must be a contiguous float64 array of the fine operator's size")
    elif (x0 is not None and np.shares_memory(out, x0)) or np.shares_memory(out, b):
        raise ArgumentError("multigrid_v_cycle: out must not alias x0 or b (they are not modified)")
    c = H.ctx
    c.check(c.lib.aggmg_vcycle(c.handle, H.handle, None if x0 is None else _pd(x0), _pd(b), int(nPre), int(nPost), float(alpha),
                               _pd(out)))
    return out


def ldiv(*args):
    """ldiv!(H, b) -- overwrites b (src/solvers.jl:63-71);  ldiv!(y, H, b) (src/solvers.jl:84-92).
    One V-cycle from a zero initial guess."""
    if len(args) == 2:
        H, b = args
        y = b
    elif len(args) == 3:
        y, H, b = args
    else:
        raise TypeError("ldiv(H, b) or ldiv(y, H, b)")
    N = H._ops[0].shape[0]
    u0 = None        # zero initial guess: aggmg_vcycle(x0 = NULL), no vector of zeros crosses PCIe
    direct = (y is not b and isinstance(y, np.ndarray) and y.dtype == np.float64 and y.shape == (N,) and y.flags["C_CONTIGUOUS"]
              and not np.shares_memory(y, np.asarray(b)))
    if direct:      # straight into y: with y and b page-locked (Context.pin) nothing is staged
        multigrid_v_cycle(H, u0, b, out=y)
    else:
        y[:] = multigrid_v_cycle(H, u0, b)
    return None


def _device_residual_norm(H_or_op, x, b):
    op = H_or_op
    r = np.empty(op.shape[0])
    c = op.ctx
    c.check(c.lib.aggmg_residual(c.handle, op.handle, _pd(_f64(x)), _pd(_f64(b)), _pd(r)))
    return np.linalg.norm(r, 2)


def norm2(x, ctx=None):
    """||x||_2 of a DeviceVector (or of a host array, uploaded first) -- C ABI aggmg_norm2_dev"""
    c = ctx or getattr(x, "ctx", None) or default_context()
    dx = x if isinstance(x, DeviceVector) else c.to_device(_f64(x))
    out = ctypes.c_double(0.0)
    c.check(c.lib.aggmg_norm2_dev(c.handle, dx.ptr, dx.n, ctypes.byref(out)))
    return out.value


def dot(x, y, ctx=None):
    """x . y of two DeviceVectors (host arrays are uploaded first) -- C ABI aggmg_dot_dev"""
    c = ctx or getattr(x, "ctx", None) or default_context()
    dx = x if isinstance(x, DeviceVector) else c.to_device(_f64(x))
    dy = y if isinstance(y, DeviceVector) else c.to_device(_f64(y))
    if dx.n != dy.n:
        raise DimensionMismatch("dot: vectors differ in length")
    out = ctypes.c_double(0.0)
    c.check(c.lib.aggmg_dot_dev(c.handle, dx.ptr, dy.ptr, dx.n, ctypes.byref(out)))
    return out.value


class DirectSolver:
    """`A \\ b` for a device operator -- the fine-level direct solve behind the reference's `err` histories
    (`u_exact = H.mStiffness[1] \\ b`, src/solvers.jl:120; `uExact = A \\ b`, :194).  A one-level hierarchy of the operator
    IS the direct solve (src/solvers.jl:39): block cyclic reduction on the device when A is block-tridiagonal with
    well-conditioned pivot blocks (every DG / agglomerated operator; accepted on a probe solve, include/aggmg_hip.h),
    the library's host banded LU otherwise; operators neither can take (CG operators in the vertices-first numbering:
    no band) are solved with SciPy's sparse LU on the host and the solution is uploaded -- once per call, not per cycle.
    `where` says which: 'device', 'host banded LU', 'host sparse LU'."""

    def __init__(self, op, host_matrix=None):
        self.op, self.ctx = op, op.ctx
        self._host = host_matrix
        self.H = None
        self._lu = None
        try:
            self.H = MeshHierarchy(None, [op], [], [], ctx=self.ctx, coarse_mode=_lib.COARSE_AUTO)
            self.where = "device" if self.H.coarse_info()["on_device"] else "host banded LU"
        except _lib.UnsupportedError:
            self.where = "host sparse LU"

    def solve_dev(self, b):
        """b: DeviceVector -> DeviceVector"""
        N = self.op.shape[0]
        if self.H is not None:
            x, z = self.ctx.alloc(N), self.ctx.alloc(N)
            self.H.vcycle_dev(z, b, x, 0, 0, 1.0)
            return x
        if self._lu is None:
            A = self._host if self._host is not None else self.op.to_scipy()
            self._lu = spla.splu(sp.csc_matrix(A))
        return self.ctx.to_device(self._lu.solve(b.download()))


def _direct_solver(owner, op, host_matrix=None):
    """the DirectSolver of `op`, cached on `owner` (a hierarchy or a smoother: factored once, reused by every call)"""
    ds = owner.__dict__.get("_direct_solver")
    if ds is None or ds.op is not op:
        ds = DirectSolver(op, host_matrix)
        owner.__dict__["_direct_solver"] = ds
    return ds


def multigrid_dev(H, x0, b, maxiter, tol, check_every=1, nPre=3, nPost=3, alpha=2.0 / 3.0, u_exact=None):
    """The loop of multigrid (src/solvers.jl:122-134) resident on the device -- C ABI
    aggmg_multigrid_dev.  x0, b: DeviceVectors.  -> (x DeviceVector, cycles, res list) or, with u_exact (DeviceVector:
    the fine-level direct solution of :120), -> (x, cycles, res list, err list) with err[i] = ||x_i - u_exact||_2 (:128)
    formed on the device."""
    c = H.ctx
    N = H._ops[0].shape[0]
    if x0.n != N or b.n != N or (u_exact is not None and u_exact.n != N):
        raise DimensionMismatch("multigrid: x0 / b do not match the fine operator")
    x = c.alloc(N)
    nchk = max(1, -(-int(maxiter) // max(1, int(check_every))))
    hist, ehist = np.zeros(nchk), np.zeros(nchk)
    ncyc, nck = ctypes.c_int(0), ctypes.c_int(0)
    c.check(c.lib.aggmg_multigrid_dev(c.handle, H.handle, x0.ptr, b.ptr, int(maxiter), float(tol), int(check_every),
                                      int(nPre), int(nPost), float(alpha), x.ptr, _pd(hist),
                                      ctypes.byref(ncyc), ctypes.byref(nck), _ptr(u_exact),
                                      _pd(ehist) if u_exact is not None else None))
    if u_exact is None:
        return x, ncyc.value, hist[:nck.value].tolist()
    return x, ncyc.value, hist[:nck.value].tolist(), ehist[:nck.value].tolist()


def multigrid(H, x0, b, maxiter, tol, exact=True, check_every=1, nPre=3, nPost=3, alpha=2.0 / 3.0):
    """multigrid(H, x0, b, maxiter, tol) -> (x, iter, res, err)  (src/solvers.jl:116-139).
    The whole loop runs on the device (aggmg_multigrid_dev): x0 and b go up once, x comes back once.  exact=True (the
    reference's contract, the default): `u_exact = H.mStiffness[1] \\ b` (:120) is solved once -- on the device where the
    fine operator is block-tridiagonal (DirectSolver) -- and err[i] = ||x_i - u_exact||_2 (:128) is formed on the device;
    exact=False skips the direct solve and returns err empty.  check_every = c > 1 runs c cycles per residual check in
    one fused device call; res / err then have one entry per check and `iter` counts cycles.  x0, b may be DeviceVectors,
    x then is one too.  nPre / nPost / alpha: multigrid_v_cycle's defaults, which the reference's loop uses (:125)."""
    c = H.ctx
    on_device = isinstance(x0, DeviceVector) and isinstance(b, DeviceVector)
    dx0 = x0 if isinstance(x0, DeviceVector) else c.to_device(_f64(x0))
    db = b if isinstance(b, DeviceVector) else c.to_device(_f64(b))
    u_exact = None
    if exact:
        A0 = H.mStiffness[0]
        u_exact = _direct_solver(H, H._ops[0], None if isinstance(A0, DeviceOperator) else A0).solve_dev(db)
    out = multigrid_dev(H, dx0, db, maxiter, tol, check_every, nPre, nPost, alpha, u_exact)
    dx, ncyc, res = out[:3]
    err = out[3] if exact else []
    if int(maxiter) == 0:
        res, err = [], []
    return (dx if on_device else dx.download()), (ncyc if check_every > 1 else len(res)), res, err


def pcg(H, b, x0=None, maxiter=50, tol=1e-10, nPre=3, nPost=3, alpha=2.0 / 3.0):
    """Conjugate gradients with ldiv!(y, H, r) (src/solvers.jl:84-92) as the preconditioner,
    resident on the device (C ABI aggmg_pcg_dev).  EXTENSION: the reference offers ldiv! for this
    use but has no Krylov loop.  -> (x, iter, res)"""
    b = _f64(b)
    c = H.ctx
    N = H._ops[0].shape[0]
    if b.shape != (N,):
        raise DimensionMismatch("pcg: b does not match the fine operator")
    dx = c.to_device(np.zeros(N) if x0 is None else _f64(x0))
    db = c.to_device(b)
    hist = np.zeros(max(1, int(maxiter)))
    it = ctypes.c_int(0)
    c.check(c.lib.aggmg_pcg_dev(c.handle, H.handle, db.ptr, dx.ptr, int(maxiter), float(tol), int(nPre), int(nPost),
                                float(alpha), _pd(hist), ctypes.byref(it)))
    return dx.download(), it.value, hist[:it.value].tolist()


def iterative_smoother_solve(A, smoother, x0, b, maxiter=1000, tol=1e-6, alpha=1.0, exact=True, check_every=1):
    """iterative_smoother_solve(A, smoother, x0, b; maxiter=1000, tol=1e-6, alpha=1.0)
    -> (x, iter, res, err)  (src/solvers.jl:189-213).  Each iteration is one fused device sweep
    x = x0 + apply_smoother(S, b - A*x0; alpha); the loop, the residual norms and -- exact=True, the reference's contract
    and the default -- the error history err[i] = ||x_i - A \\ b||_2 (:194, :202) stay on the device
    (aggmg_smoother_solve_dev; the direct solve once, DirectSolver).  exact=False: err comes back empty."""
    op = smoother.A if not isinstance(A, DeviceOperator) else A
    c = op.ctx
    dx0, db = c.to_device(_f64(x0)), c.to_device(_f64(b))
    u_exact = None
    if exact:
        u_exact = _direct_solver(smoother, op, None if isinstance(A, DeviceOperator) else A).solve_dev(db)
    dx, nit, res, err = smoother_solve_dev(op, smoother, dx0, db, maxiter, tol, alpha, check_every=check_every, u_exact=u_exact)
    return dx.download(), (nit if check_every > 1 else len(res)), res, err


def smoother_solve_dev(A_op, smoother, x0, b, maxiter, tol, alpha=1.0, check_every=1, u_exact=None):
    """The loop of iterative_smoother_solve (src/solvers.jl:196-208) resident on the device -- C ABI
    aggmg_smoother_solve_dev.  x0, b (and u_exact, the direct solution of :194, or None): DeviceVectors.
    -> (x DeviceVector, iterations, res list, err list ([] without u_exact))."""
    c = A_op.ctx
    N = A_op.shape[0]
    if x0.n != N or b.n != N or (u_exact is not None and u_exact.n != N):
        raise DimensionMismatch("iterative_smoother_solve: x0 / b do not match the operator")
    dx = c.alloc(N)
    nchk = max(1, -(-int(maxiter) // max(1, int(check_every))))
    hist, ehist = np.zeros(nchk), np.zeros(nchk)
    nit, nck = ctypes.c_int(0), ctypes.c_int(0)
    c.check(c.lib.aggmg_smoother_solve_dev(c.handle, A_op.handle, smoother.handle, x0.ptr, b.ptr, int(maxiter),
                                           float(tol), float(alpha), int(check_every), dx.ptr, _pd(hist),
                                           ctypes.byref(nit), ctypes.byref(nck), _ptr(u_exact),
                                           _pd(ehist) if u_exact is not None else None))
    res = hist[:nck.value].tolist()
    err = ehist[:nck.value].tolist() if u_exact is not None else []
    return dx, nit.value, res, err


# stand-alone fused operations on host arrays (C ABI `aggmg_smooth`, `aggmg_residual`, ...)
def smooth(A_op, S, u, b, alpha=2.0 / 3.0, nsweeps=1):
    """nsweeps x `u += apply_smoother(S, b - A*u; alpha)` (src/solvers.jl:32-35) -> new vector"""
    u = _f64(u).copy()
    b = _f64(b)
    c = A_op.ctx
    c.check(c.lib.aggmg_smooth(c.handle, A_op.handle, S.handle, _pd(u), _pd(b), float(alpha), int(nsweeps)))
    return u


def residual(A_op, u, b):
    """b - A*u (src/solvers.jl:33)"""
    r = np.empty(A_op.shape[0])
    c = A_op.ctx
    c.check(c.lib.aggmg_residual(c.handle, A_op.handle, _pd(_f64(u)), _pd(_f64(b)), _pd(r)))
    return r


def restrict(L_op, r):
    """L' * r (src/solvers.jl:36)"""
    rc = np.empty(L_op.shape[1])
    c = L_op.ctx
    c.check(c.lib.aggmg_restrict(c.handle, L_op.handle, _pd(_f64(r)), _pd(rc)))
    return rc


def prolong_add(L_op, uc, u):
    """u + L * uc (src/solvers.jl:42) -> new vector"""
    u = _f64(u).copy()
    c = L_op.ctx
    c.check(c.lib.aggmg_prolong_add(c.handle, L_op.handle, _pd(_f64(uc)), _pd(u)))
    return u


def smoother_launch_bytes(op, smoother=None, what="sweeps"):
    """(read, write) compulsory HBM bytes of one aggmg_smooth_dev launch (what='sweeps') or one aggmg_residual_dev
    launch (what='residual') on operator `op` (C ABI aggmg_smoother_launch_bytes): every array counted once."""
    ctx = op.ctx
    r, w = ctypes.c_int64(0), ctypes.c_int64(0)
    ctx.check(ctx.lib.aggmg_smoother_launch_bytes(ctx.handle, op.handle, smoother.handle if smoother is not None else None,
                                                  {"sweeps": 0, "residual": 1}[what], ctypes.byref(r), ctypes.byref(w)))
    return r.value, w.value
